"""Empty stand-in for `assimulo` (absent). The Michaelis-Menten scripts import
it at top level but never call it; the methanation time integration is NOT
made runnable by this (it stays parity-unpinned)."""
