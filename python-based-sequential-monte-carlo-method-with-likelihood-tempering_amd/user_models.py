"""Example sources for HipEngine.set_model_user (include/smc_hip.h: smc_set_model_user) - what a user of the
reference writes instead of a new Micmem_likelihood.py (README.md:4, "modify for your problem")."""

# Micmem_likelihood.py:14-33 as a user model: theta = (Vmax, Km, sigma), cond = (S0,), one state S, observed P = S0 - S
MICHAELIS_MENTEN_PLAIN = r"""
__device__ void smc_user_y0(const double *theta, const double *cond, double *y) { y[0] = cond[0]; }
__device__ void smc_user_rhs(double t, const double *y, const double *theta, const double *cond, double *dydt) {
    dydt[0] = smc_div((-theta[0]) * y[0], theta[1] + y[0]);     // a / b, faster (include/smc_hip.h)
}
__device__ double smc_user_obs(double t, const double *y, const double *theta, const double *cond) { return cond[0] - y[0]; }
"""
# ... with the optional cost hint (include/smc_hip.h): explicit RK45 runs on its stability limit for about 3.7 Vmax / Km step
# attempts, so the sweeps hand those solves out first and run the longest one per wave - results are the same either way
MICHAELIS_MENTEN = MICHAELIS_MENTEN_PLAIN + r"""
__device__ double smc_user_cost(const double *theta) { return theta[1] > 0.0 ? 3.7 * theta[0] / theta[1] : 0.0; }
"""

# two states: A -> B -> C with rate constants theta = (k1, k2, sigma), cond = (A0,), the intermediate B is observed
CONSECUTIVE_REACTIONS = r"""
__device__ void smc_user_y0(const double *theta, const double *cond, double *y) { y[0] = cond[0]; y[1] = 0.0; }
__device__ void smc_user_rhs(double t, const double *y, const double *theta, const double *cond, double *dydt) {
    dydt[0] = -theta[0] * y[0];
    dydt[1] = theta[0] * y[0] - theta[1] * y[1];
}
__device__ double smc_user_obs(double t, const double *y, const double *theta, const double *cond) { return y[1]; }
"""
