"""Device-side counterpart of the reference's NumPy frequency-domain backend (PC/application/realtime_scripts)."""
