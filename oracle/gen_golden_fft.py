#!/usr/bin/env python3
"""Golden vectors for the frequency-domain (phase-steer) beamformer, from the REAL reference.

TEST INFRASTRUCTURE ONLY.  PC/application/realtime_scripts is pure NumPy, so it is imported as it lies (cwd =
/root/reference/PC/application, nothing copied) and fed synthetic blocks; inputs and outputs are stored as data in
tests/golden/fft_backend.npz.  Run in the build container:  python oracle/gen_golden_fft.py"""
import os
import subprocess
import sys
import textwrap

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
APP = "/root/reference/PC/application"

WORKER = textwrap.dedent('''
    import sys, hashlib
    import numpy as np
    import matplotlib
    matplotlib.use("Agg")
    sys.path.insert(0, "")
    out_path = sys.argv[1]
    import realtime_scripts.config as config
    import realtime_scripts.calc_phase_shift_cartesian as cps
    import realtime_scripts.beam_forming_algorithm as bfa
    out = {}
    out["active_mics"] = np.asarray(cps.active_mics, dtype=np.int64)
    out["r_prime_all"] = np.asarray(cps.r_prime_all)
    out["x_scan"] = np.asarray(cps.x_scan).ravel(); out["y_scan"] = np.asarray(cps.y_scan).ravel()
    out["freq"] = np.asarray(cps.f).ravel()
    out["bin_lo"] = np.array(int(cps.threshold_freq_lower_idx)); out["bin_hi"] = np.array(int(cps.threshold_freq_upper_idx))
    ph = np.asarray(cps.phase_shift)
    out["phase_shift_shape"] = np.array(ph.shape)
    out["phase_shift_sha256"] = np.array(hashlib.sha256(np.ascontiguousarray(ph).tobytes()).hexdigest())
    out["phase_shift_probe"] = ph[::9, ::17, ::3, ::4].copy()
    out["cfg"] = np.array([config.N_SAMPLES, config.MAX_RES_X, config.MAX_RES_Y, config.ACTIVE_ARRAYS, config.N_MICROPHONES], dtype=np.int64)
    out["cfg_f"] = np.array([config.PROPAGATION_SPEED, float(config.fs), config.ELEMENT_DISTANCE, config.VIEW_ANGLE, config.Z, config.ASPECT_RATIO,
                             config.threshold_freq_lower, config.threshold_freq_upper], dtype=np.float64)
    n_act = len(cps.active_mics)
    N = config.N_SAMPLES
    rng = np.random.default_rng(11)
    t = np.arange(N) / float(config.fs)
    # F1: white noise; F2: plane wave from scan point (x 9, y 4): mic i sees s(t + tau_i), tau from the reference's own geometry
    sigs = {"f1": rng.standard_normal((N, n_act)) * 0.1}
    xs, ys = out["x_scan"][9], out["y_scan"][4]
    r = np.sqrt(xs * xs + ys * ys + config.Z ** 2)
    xi = cps.r_prime_all[0, cps.active_mics]; yi = cps.r_prime_all[1, cps.active_mics]
    tau = (xs * xi + ys * yi) / r / config.PROPAGATION_SPEED
    s2 = np.zeros((N, n_act))
    for fr in (2000.0, 4500.0, 9000.0):
        s2 += np.sin(2 * np.pi * fr * (t[:, None] + tau[None, :]))
    sigs["f2"] = s2 + rng.standard_normal((N, n_act)) * 0.05
    sigs["f3"] = sigs["f1"] * 1e-3          # below threshold_heatmap: the reference returns zeros
    for k, v in sigs.items():
        v32 = v.astype(np.float32)
        out["in_" + k] = v32
        shifted = bfa.frequency_phase_shift(v32, cps.phase_shift)
        power = np.sum((np.abs(np.sum(shifted, axis=1))) ** 2, axis=0)
        out["power_" + k] = power                      # before normalisation (float64 [X, Y])
        out["heatmap_" + k] = np.array(bfa.main(v32))  # what the reference returns
    np.savez_compressed(out_path, **out)
    print("wrote", out_path, "active", n_act, "bins", out["bin_lo"], out["bin_hi"], "argmax f2", np.unravel_index(np.argmax(out["power_f2"]), out["power_f2"].shape))
''')


def main():
    out = os.path.join(REPO, "tests", "golden", "fft_backend.npz")
    worker = "/tmp/golden_fft_worker.py"
    open(worker, "w").write(WORKER)
    subprocess.check_call([sys.executable, worker, out], cwd=APP)


if __name__ == "__main__":
    main()
