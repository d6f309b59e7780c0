"""Constants of the reference's frequency-domain backend (PC/application/realtime_scripts/config.py:1-52; it keeps its
own hand-edited copy, different from PC/src/config.json).  Plain module attributes so callers can change them before
building a FrequencyBeamformer."""
N_MICROPHONES = 256
N_SAMPLES = 256
MAX_RES_X = 13
MAX_RES_Y = 13
Z = 1.0
VIEW_ANGLE = 68.0
ELEMENT_DISTANCE = 0.02
ARRAY_SEPARATION = 0.0
ACTIVE_ARRAYS = 4
PROPAGATION_SPEED = 343.0
ASPECT_RATIO = 16 / 9
SAMPLE_RATE = 48828.0
columns = 8
rows = 8
mode = 1
threshold_freq_lower = 0
threshold_freq_upper = 18000
fs = int(48828)
CAMERA_OFFSET = 0.11        # calc_r_prime.py:7
