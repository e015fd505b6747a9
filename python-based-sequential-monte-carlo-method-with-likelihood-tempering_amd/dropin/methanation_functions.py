"""Shadow of SMC_methanation/methanation_functions.py: the likelihood fan-out (`sim_particle`, `cal_parallel_new`,
:44-92), the prior density (`cal_prior`, :96-135) and the five output helpers the driver calls
(`SMC_methanation_main.py:185,199,421-438`; reference :139-272) so that the UNMODIFIED driver runs on these modules
(tests/test_reference_meth_driver_on_shadows.py).  One batched GPU call replaces the one-Ray-task-per-particle
fan-out.  The CSV helper writes the reference's two files through the engine's dump code
(`smc_lt_amd.driver.write_particles_csv`); the four figure helpers are deliberately plain (figures are outside the hot
path, SURVEY.md section 1 row Lp): same call signatures, same output file names, one figure per call, and they never
abort a run - without matplotlib they say so and return."""
import numpy as np
import scipy.stats

from methanation_set_conditon import *  # noqa: F401,F403
from methanation_set_likelihood import *  # noqa: F401,F403
from methanation_set_likelihood import my_model_batch as _my_model_batch
from smc_lt_amd import methanation as _gpu


def cal_parallel_new(params, obs_data, initial_guess):
    """(lk, molfraction) for one 9-vector (8 kinetic parameters + sigma), :44-65."""
    params = np.asarray(params, dtype=np.float64)
    sigma = params[-1] if est_sigma else sigma_true   # noqa: F405
    ycal1, molfraction = my_model(params, initial_guess)   # noqa: F405
    lk = my_loglike(ycal1, obs_data, sigma, n_data, scale=1.0)   # noqa: F405
    return lk, molfraction


cal_parallel_new.remote = cal_parallel_new


def sim_particle(particle, initial_guess, obs_data, p_pred_bases):
    """:70-92 - scatter the estimated columns into the base rows, evaluate every particle, return (llk, C_l_)."""
    print('sim_particle')
    p_pred_bases[:, est_position] = particle   # noqa: F405
    rows = np.ascontiguousarray(p_pred_bases[:len(particle)], dtype=np.float64)
    Flow, mol = _my_model_batch(rows, initial_guess)
    sigma = rows[:, -1] if est_sigma else np.full(len(rows), float(sigma_true))   # noqa: F405
    llk = _gpu.my_loglike(Flow, np.asarray(obs_data, dtype=np.float64), sigma, n_data)   # noqa: F405
    return llk, list(mol)


def cal_prior(theta):
    """:96-135, uniform branch (normal_pred = False is the reference's setting; the Gaussian branches are dead)."""
    if normal_pred:   # noqa: F405
        raise NotImplementedError("normal_pred=True is dead code in the reference (methanation_set_conditon.py:23)")
    dl = high_limit_array - low_limit_array   # noqa: F405
    p = scipy.stats.uniform.pdf(theta, [low_limit[i] for i in est_position], dl)   # noqa: F405
    return np.prod(p.T, axis=0)


# ---------------------------------------------------------------------------------------------------------------
# output helpers of the driver (reference :139-272).  Names, argument order and file names are the boundary; what is
# drawn is this package's own minimal rendering.
# ---------------------------------------------------------------------------------------------------------------
_PARAM_LABELS = ("Af", "Eaf", "Ar", "Ear", "BCO2", "dHCO2", "BH2O", "dHH2O", "sigma")


def _pyplot():
    try:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        return plt
    except Exception as exc:   # no matplotlib (or no usable backend): figures are skipped, the run goes on
        print(f"[smc_lt_amd] figure skipped: {exc.__class__.__name__}: {exc}")
        return None


def _stacked_histograms(plt, columns, labels, path, marks=None, bins=50, ranges=None):
    """One row per estimated parameter; `columns` is a list of (values, colour) lists per row."""
    fig, axes = plt.subplots(len(columns), 1, figsize=(8, 2.2 * len(columns)), squeeze=False)
    for k, (ax, sets) in enumerate(zip(axes[:, 0], columns)):
        for vals, colour in sets:
            vals = np.asarray(vals, dtype=np.float64)
            vals = vals[np.isfinite(vals)]
            rng_k = None if ranges is None else ranges[k]
            if vals.size:
                ax.hist(vals, bins, range=rng_k, density=True, color=colour)
                ax.axvline(vals.mean(), color=colour[:3], linestyle="dashed", linewidth=1)
        if marks is not None:
            ax.axvline(marks[k], color="black", linewidth=2)
        ax.set_ylabel(labels[k])
        ax.grid(True)
    fig.tight_layout()
    fig.savefig(path, dpi=120)
    plt.close(fig)


def _est_labels():
    return [_PARAM_LABELS[i] for i in est_position]   # noqa: F405


def DistributionDrawerWhileSMC(p_filt, dirname, name_):
    """Histogram of every estimated parameter over its prior box, with the generating value marked (:190-205).
    Called for the prior, after every tempering step and for the posterior (SMC_methanation_main.py:185,423,432)."""
    plt = _pyplot()
    if plt is None:
        return
    p_filt = np.asarray(p_filt, dtype=np.float64)
    d = p_filt.shape[1]
    box = [(float(low_limit_array[k]), float(high_limit_array[k])) for k in range(d)]   # noqa: F405
    marks = [float(baseparams_withsigma[i]) for i in est_position][:d]   # noqa: F405
    _stacked_histograms(plt, [[(p_filt[:, k], (0.2, 0.4, 0.8, 1.0))] for k in range(d)], _est_labels()[:d],
                        f"{dirname}{name_}.png" if "." not in str(name_) else f"{dirname}{name_}", marks=marks, ranges=box)


def ParityplotDrawerWhileSMC(obs_data, dirname01_, dirname02, C, name_):
    """Simulated outlet mole fraction of every particle against the observation, one figure per species and per output
    directory (:140-187; SMC_methanation_main.py:199,421).  C is sim_particle's second return value: per particle a
    (5, n_data) array."""
    plt = _pyplot()
    if plt is None:
        return
    obs_data = np.asarray(obs_data, dtype=np.float64)
    sim = np.asarray([np.asarray(c, dtype=np.float64)[:5] for c in C])           # (n_particle, 5, n_data)
    for i in range(5):
        x = obs_data[i, :sim.shape[2]]
        q25, q50, q75 = np.percentile(sim[:, i, :], [25, 50, 75], axis=0)
        mean = sim[:, i, :].mean(axis=0)
        for folder, centre, lo, hi in ((dirname01_, q50, q25, q75), (dirname02, mean, None, None)):
            fig, ax = plt.subplots(figsize=(5, 5))
            ax.plot([0, 1], [0, 1], "r--")
            if lo is None:
                ax.plot(x, centre, "o")
            else:
                ax.errorbar(x, centre, yerr=[np.maximum(centre - lo, 0), np.maximum(hi - centre, 0)], fmt="o", capsize=2)
            ax.set_xlabel(f"data X{'abcde'[i]} [-]")
            ax.set_ylabel(f"simulation X{'abcde'[i]} [-]")
            ax.set_ylim(-0.05, 1)
            fig.savefig(f"{folder}Overlayed_Simulation_while_SMC_{name_}_N_{i}.png", dpi=100)
            plt.close(fig)


def SavePosteriorPairplot(p_filt, dirname, name):
    """Lower-triangle scatter matrix of the posterior sample with marginal histograms on the diagonal (:208-227; the
    reference uses seaborn.pairplot, which this module does not need)."""
    plt = _pyplot()
    if plt is None:
        return
    p_filt = np.asarray(p_filt, dtype=np.float64)
    d = p_filt.shape[1]
    labels = _est_labels()[:d]
    fig, axes = plt.subplots(d, d, figsize=(2.2 * d, 2.2 * d), squeeze=False)
    for r in range(d):
        for c in range(d):
            ax = axes[r, c]
            if c > r:
                ax.set_visible(False)
            elif c == r:
                ax.hist(p_filt[:, r], 40, density=True)
            else:
                ax.plot(p_filt[:, c], p_filt[:, r], ".", markersize=2)
            if r == d - 1:
                ax.set_xlabel(labels[c])
            if c == 0:
                ax.set_ylabel(labels[r])
    fig.tight_layout()
    fig.savefig(f"{dirname}{name}.png", dpi=100)
    plt.close(fig)


def SavePosteriorcsv(p_filt, dirname, dirnamepred, name1, name2):
    """The run's two result files (:229-236): `{dirname}{name1}.csv` with a header row - the first
    num_est_params - 1 names of the parameter list, then 'sigma', exactly as the reference labels them - and the bare
    `{dirnamepred}{name2}.csv` that ComparePriorPosterior and a resumed run read back."""
    from smc_lt_amd.driver import write_particles_csv
    header = [_PARAM_LABELS[i] for i in range(num_est_params - 1)] + ["sigma"]   # noqa: F405
    write_particles_csv(f"{dirname}{name1}.csv", np.asarray(p_filt), header)
    write_particles_csv(f"{dirnamepred}{name2}.csv", np.asarray(p_filt))


def ComparePriorPosterior(firstpred, last_pred, dirname, name):
    """Prior and posterior sample of every estimated parameter over their common range (:238-272); both arguments are
    paths of bare CSV dumps (pred/first_p_pred.csv, pred/last_p_pred.csv)."""
    a = np.atleast_2d(np.loadtxt(firstpred, delimiter=","))
    b = np.atleast_2d(np.loadtxt(last_pred, delimiter=","))
    plt = _pyplot()
    if plt is None:
        return
    d = min(a.shape[1], b.shape[1])
    ranges = [(float(min(a[:, k].min(), b[:, k].min())), float(max(a[:, k].max(), b[:, k].max()))) for k in range(d)]
    marks = [float(baseparams_withsigma[i]) for i in est_position][:d]   # noqa: F405
    _stacked_histograms(plt, [[(a[:, k], (0.0, 0.0, 1.0, 0.3)), (b[:, k], (1.0, 0.0, 0.0, 0.7))] for k in range(d)],
                        _est_labels()[:d], f"{dirname}{name}.png", marks=marks, bins=n_hist, ranges=ranges)   # noqa: F405
