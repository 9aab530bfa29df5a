"""Lane-level emulation of k_field / k_light (csrc/dsn_field.hip) in numpy on the HOST image produced
by the library's own packing code (dsn_pack_params_host_image): checks the MFMA operand permutation,
the linear weight stream and the register chaining against the reference goldens without a GPU."""
import ctypes as C

import numpy as np
import pytest

import oracle as O
from helpers import load, state

BLK = 1024
lanes = np.arange(64)
half = lanes >> 5
COL = lanes & 31


def crow(r, h):
    return (r & 3) + 8 * (r >> 2) + 4 * h


ROW = np.array([[crow(r, h) for h in half] for r in range(16)])


def mfma(A, B, acc):   # v_mfma_f32_32x32x2_f32: lane l gives A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]
    D = A.reshape(2, 32).T.astype(np.float64) @ B.reshape(2, 32).astype(np.float64)
    acc += D[ROW, COL[None, :]]


@pytest.fixture(scope="module")
def packed():
    import dsnerf_amd
    lib = dsnerf_amd._lib.lib()
    sd = state()
    P = O.Params(sd)
    buf = np.zeros(lib.dsn_packed_param_bytes() // 4, np.float32)
    assert lib.dsn_pack_params_host_image(P.ptrs, buf.ctypes.data_as(C.c_void_p)) == 0
    return sd, buf


def offsets():
    names = ["L0", "L1", "L2", "L3", "L4", "L5", "L6", "RGB1", "L6T", "L5T", "L4T", "L3T", "L2T", "L1T", "L0T", "LT0", "LT1"]
    sizes = [16, 64, 64, 64, 80, 64, 64, 32, 64, 64, 80, 64, 64, 64, 16, 4, 16]
    off, o = {}, 0
    for n, s in zip(names, sizes):
        off[n] = o
        o += s * BLK
    for n, s in [("B1", 6 * 256), ("BRGB1", 128), ("WDEN", 256), ("WRGB3", 384), ("BLT0", 128), ("BLT1", 128),
                 ("WLT2", 128), ("SCAL", 64)]:
        off[n] = o
        o += s
    return off


def test_field_and_light_dataflow(packed):
    sd, pk = packed
    OFF = offsets()
    g = load("full_eval")
    act = np.nonzero(~g["transparent"])[0][:32]
    xc = g["x_c"][act].astype(np.float64)
    code = sd["nerf.embedding.weight"][int(g["frame"])]
    W0 = sd["nerf.stage1.0.weight"]
    bias0 = sd["nerf.stage1.0.bias"] + W0[:, :8] @ code + W0[:, 71:87] @ g["pose_feat"][0]

    pos = [0]

    def dense(inregs, acc):
        for kb in range(inregs.shape[0]):
            blk = pk[pos[0]:pos[0] + BLK].reshape(4, 64, 4)
            for r in range(16):
                mfma(blk[r // 4, :, r % 4], inregs[kb][r], acc)
            pos[0] += BLK

    def rows(off, m):
        return np.array([[pk[off + 32 * m + crow(r, h)] for h in half] for r in range(16)], np.float64)

    x = xc[COL]
    pe = np.zeros((2, 16, 64))
    for t in range(30):
        j, a = t // 3, t % 3
        arg = x[:, a] * (1 << j)
        pe[t >> 4][t & 15] = np.where(half == 1, np.cos(arg), np.sin(arg))
    pe[1][14] = np.where(half == 1, x[:, 1], x[:, 0])
    pe[1][15] = np.where(half == 1, 0, x[:, 2])

    masks = []
    pos[0] = OFF["L0"]
    h = np.zeros((8, 16, 64))
    mk = np.zeros((8, 16, 64), bool)
    for m in range(8):
        acc = np.array([[bias0[32 * m + crow(r, hh)] for hh in half] for r in range(16)], np.float64)
        dense(pe, acc)
        mk[m] = acc > 0
        h[m] = np.maximum(acc, 0)
    masks.append(mk)

    def layer_fwd(hin, boff, extra=None):
        out = np.zeros((8, 16, 64))
        mk = np.zeros((8, 16, 64), bool)
        for m in range(8):
            acc = rows(boff, m)
            dense(hin, acc)
            if extra is not None:
                dense(extra, acc)
            mk[m] = acc > 0
            out[m] = np.maximum(acc, 0)
        masks.append(mk)
        return out

    for l in range(3):
        h = layer_fwd(h, OFF["B1"] + l * 256)
    h = layer_fwd(h, OFF["B1"] + 3 * 256, extra=pe)
    h = layer_fwd(h, OFF["B1"] + 4 * 256)
    h = layer_fwd(h, OFF["B1"] + 5 * 256)
    part = sum((rows(OFF["WDEN"], m) * h[m]).sum(0) for m in range(8))
    sig = part[:32] + part[32:] + pk[OFF["SCAL"]]
    e = np.zeros((3, 64))
    for m in range(4):
        acc = rows(OFF["BRGB1"], m)
        dense(h, acc)
        v = np.maximum(acc, 0)
        for c in range(3):
            e[c] += (rows(OFF["WRGB3"] + c * 128, m) * v).sum(0)
    ess = (e[:, :32] + e[:, 32:]).T + pk[OFF["SCAL"] + 1:OFF["SCAL"] + 4]
    assert pos[0] == OFF["L6T"]        # the stream is linear: heads end where the reverse pass starts

    gA = np.array([rows(OFF["WDEN"], m) * masks[6][m] for m in range(8)])

    def layer_bwd(gin, mk):
        out = np.zeros((8, 16, 64))
        for m in range(8):
            acc = np.zeros((16, 64))
            dense(gin, acc)
            out[m] = acc * mk[m]
        return out

    gB = layer_bwd(gA, masks[5])
    gA = layer_bwd(gB, masks[4])
    gB = layer_bwd(gA, masks[3])
    dpe = np.zeros((2, 16, 64))
    for b in range(2):
        dense(gA, dpe[b])
    gA = layer_bwd(gB, masks[2])
    gB = layer_bwd(gA, masks[1])
    gA = layer_bwd(gB, masks[0])
    for b in range(2):
        dense(gA, dpe[b])
    assert pos[0] == OFF["LT0"]
    gr = np.zeros((3, 64))
    for t in range(30):
        j, a = t // 3, t % 3
        term = dpe[t >> 4][t & 15] * pe[t >> 4][t & 15][lanes ^ 32] * (1 << j)
        gr[a] += np.where(half == 1, -term, term)
    gr[1] += np.where(half == 1, dpe[1][14], 0)
    gr[0] += np.where(half == 0, dpe[1][14], 0)
    gr[2] += np.where(half == 0, dpe[1][15], 0)
    grad = (gr[:, :32] + gr[:, 32:]).T

    assert np.abs(sig - g["sigma"][act]).max() < 5e-5
    assert np.abs(ess - g["essence"][act]).max() < 5e-6
    assert np.abs(grad - g["grad_sigma"][act]).max() < 2e-5 * np.abs(g["grad_sigma"][act]).max()

    # ---- lighting
    S = int(g["S"])
    d = g["ray_d"][act // S]
    vd = d / np.linalg.norm(d, axis=-1, keepdims=True)
    in9 = np.concatenate([g["n_w"][act], g["pts"].reshape(-1, 3)[act], vd, np.zeros((32, 1))], 1)[COL]
    h1 = np.zeros((4, 16, 64))
    for m in range(4):
        acc = rows(OFF["BLT0"], m)
        blk = pk[OFF["LT0"] + m * BLK:OFF["LT0"] + (m + 1) * BLK].reshape(4, 64, 4)
        for r in range(5):
            mfma(blk[r // 4, :, r % 4], np.where(half == 1, in9[:, 2 * r + 1], in9[:, 2 * r]), acc)
        h1[m] = np.maximum(acc, 0)
    pos[0] = OFF["LT1"]
    h2 = np.zeros((4, 16, 64))
    for m in range(4):
        acc = rows(OFF["BLT1"], m)
        dense(h1, acc)
        h2[m] = np.maximum(acc, 0)
    part = sum((rows(OFF["WLT2"], m) * h2[m]).sum(0) for m in range(4))
    o = part[:32] + part[32:] + pk[OFF["SCAL"] + 4]
    w = np.where(o > 0, o, np.expm1(o)) + 1
    col = w[:, None] * g["essence"][act]
    assert np.abs(col - g["colour"][act]).max() < 1e-6
