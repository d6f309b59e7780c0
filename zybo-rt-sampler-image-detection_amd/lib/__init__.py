"""Python mirror of the reference's `lib` package (PC/lib/{beamformer,directions,tests}.so built by PC/setup.py)."""
