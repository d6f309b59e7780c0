"""Ingest: FPGA protocol-v2 datagrams -> mic-major float frame (PC/src/receiver.c:94-151).

CPU: the oracle restatement against the REAL receiver (compiled from the reference by oracle/build_ref.py) fed through a
UDP loopback socket, and against a committed golden hash.  GPU: the HIP transpose kernel against the oracle, bit for bit."""
import ctypes as C
import hashlib
import json
import os
import socket
import threading
import time

import numpy as np
import pytest

import util

N, M = 256, 256
STRIDE = 8 + 4 * M
GOLD = os.path.join(util.GOLDEN, "ingest_shipped.json")


def datagrams(seed=42):
    rng = np.random.default_rng(seed)
    pk = np.zeros((N, STRIDE), dtype=np.uint8)
    stream = rng.integers(-2 ** 23, 2 ** 23, size=(N, M), dtype=np.int32)
    stream[0, :8] = [2 ** 31 - 1, -2 ** 31, 16777217, -16777217, 33554433, 1, -1, 0]   # int -> float roundings
    pk[:, 8:] = stream.view(np.uint8).reshape(N, 4 * M)
    pk[:, 2], pk[:, 3] = 4, 2                                                            # n_arrays, protocol version
    return pk


def oracle_frame(oracle_lib, pk, n_arrays):
    lib = C.CDLL(os.path.join(util.ROOT, "oracle", "libdas_oracle.so"))
    out = np.zeros(M * N, dtype=np.float32)
    lib.oracle_ingest(pk.ctypes.data_as(C.c_void_p), N, M, n_arrays, 8, 8, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def test_oracle_matches_golden_hash(oracle_lib):
    gold = json.load(open(GOLD))
    pk = datagrams(gold["seed"])
    assert hashlib.sha256(pk.tobytes()).hexdigest() == gold["packets_sha256"]
    for n_arrays in (1, 3):
        out = oracle_frame(oracle_lib, pk, n_arrays)
        assert hashlib.sha256(out.tobytes()).hexdigest() == gold["frame_sha256"][str(n_arrays)]


def test_oracle_matches_real_receiver_over_udp(oracle_lib):
    path = os.path.join(util.ROOT, "oracle", "_ref", "libref_receiver_shipped.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    ref = C.CDLL(path)
    ref.create_msg.restype = C.c_void_p
    fd = ref.create_and_bind_socket(C.c_bool(True))          # 127.0.0.1:21844 (UDP_REPLAY_IP, config.json:28)
    if fd < 0:
        pytest.skip("cannot bind the replay socket")
    pk = datagrams()
    tx = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)

    def send():
        time.sleep(0.2)
        for i in range(N):
            tx.sendto(pk[i].tobytes(), ("127.0.0.1", 21844))
            if i % 16 == 15:
                time.sleep(0.002)

    th = threading.Thread(target=send)
    th.start()
    got = np.zeros(M * N, dtype=np.float32)
    ref.receive_to_buffer(C.c_int(fd), got.ctypes.data_as(C.POINTER(C.c_float)), C.c_void_p(ref.create_msg()), C.c_int(3))
    th.join()
    ref.close_socket(C.c_int(fd))
    assert np.array_equal(got, oracle_frame(oracle_lib, pk, 3))


@pytest.mark.gpu
@pytest.mark.parametrize("n_arrays", [1, 3, 4])
def test_gpu_ingest_bit_exact(native, oracle_lib, n_arrays):
    util.configure("shipped")
    pk = datagrams()
    got = np.zeros(n_arrays * 64 * N, dtype=np.float32)
    assert native.lib.bf_ingest(pk.ctypes.data_as(C.c_void_p), n_arrays, 8, 8, native.fptr(got)) == 0
    native.check()
    want = oracle_frame(oracle_lib, pk, n_arrays)[: got.size]
    assert got.tobytes() == want.tobytes()


@pytest.mark.gpu
def test_gpu_ingest_feeds_beamformer(native, oracle_lib):
    """datagrams -> frame -> mimo_pad, all on the device, equals the oracle chain."""
    import torch
    c = util.configure("shipped")
    pk = datagrams(7)
    d_pk = torch.from_numpy(pk).cuda()
    d_frame = torch.zeros((M, N), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    assert native.lib.bf_ingest_device(d_pk.data_ptr(), 3, 8, 8, d_frame.data_ptr(), stream) == 0
    whole = util.table_for("pad", "shipped").ravel()
    native.lib.load_coefficients_pad(native.iptr(whole), whole.size); native.check()
    mics = np.arange(192, dtype=np.int32)
    D = c["X"] * c["Y"]
    # table was generated for 256 mics; steer with the first 192 rows of each direction
    sub = np.ascontiguousarray(util.table_for("pad", "shipped").reshape(D, 256)[:, :192]).ravel()
    native.lib.load_coefficients_pad(native.iptr(sub), sub.size); native.check()
    d_img = torch.zeros((1, D), dtype=torch.float32, device="cuda")
    assert native.lib.bf_das_device(native.PAD, d_frame.data_ptr(), M, d_img.data_ptr(), D, 1, native.iptr(mics), 192, 0, D, stream) == 0
    torch.cuda.synchronize()
    frame = oracle_frame(oracle_lib, pk, 3).reshape(M, N)
    orc = oracle_lib.Oracle(N, c["X"], c["Y"], c["T"])
    orc.load(0, sub)
    want = orc.mimo_range(0, frame, mics, 0, D)
    assert d_img.cpu().numpy().ravel().tobytes() == want.tobytes()
