"""Inert stand-in for `seaborn` (absent; only plot helpers use it)."""
