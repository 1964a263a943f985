"""Second, independent restatement (numpy float32, vectorised) of the reference's hit shading,
used only to cross-check oracle.cpp (TEST INFRASTRUCTURE).

Follows shaders/raytrace.rchit:81-219, shaders/gltf.glsl:55-154 and shaders/random.glsl:22-70
directly; written from the GLSL, not from oracle.cpp.  Uses numpy's own sin/cos/power, so it
agrees with the oracle to rounding (a few ulp on well-conditioned inputs), not bit for bit.

Record layout = oracle.cpp orc_eval_shade: 40 floats in, 20 floats out.
"""
import numpy as np

F = np.float32
PI = F(3.14159265)
INV_PI = F(1.0) / PI


def _lcg(seed):
    seed = (np.uint32(1664525) * seed + np.uint32(1013904223)).astype(np.uint32)
    return seed, (seed & np.uint32(0x00FFFFFF))


def _rnd(seed):
    seed, bits = _lcg(seed)
    return seed, bits.astype(F) / F(16777216.0)


def _dot(a, b):
    return (a[:, 0] * b[:, 0] + a[:, 1] * b[:, 1]) + a[:, 2] * b[:, 2]


def _norm(a):
    return a * (F(1.0) / np.sqrt(_dot(a, a)))[:, None]


def _ndf(N, H, alpha):
    a2 = alpha * alpha
    NH = _dot(N, H)
    d = NH * NH * (a2 - F(1)) + F(1)
    return np.where(NH <= 0, F(0), a2 * INV_PI / (d * d + F(1e-4)))


def _g1(x, k):
    return x / (x * (F(1) - k) + k)


def _smith(N, V, L, k):
    return _g1(np.abs(_dot(N, V)), k) * _g1(np.abs(_dot(N, L)), k)


def _schlick(H, V, F0):
    return F0 + (F(1) - F0) * np.power(F(1) - np.abs(_dot(H, V)), F(5.0))[:, None]


def shade(rec):
    rec = np.ascontiguousarray(rec, F).reshape(-1, 40)
    n = rec.shape[0]
    P, Nrm, T, B, D = rec[:, 0:3], rec[:, 3:6], rec[:, 6:9], rec[:, 9:12], rec[:, 12:15]
    base, metal_f, rough_f, emis = rec[:, 15:18], rec[:, 19], rec[:, 20], rec[:, 21:24]
    lpos, lcol, lint = rec[:, 24:27], rec[:, 27:30], rec[:, 30]
    ltype = rec[:, 31].copy().view(np.int32)
    bits = rec[:, 32:36].copy().view(np.uint32)
    seed, depth, isspec_in, lcount = bits[:, 0].copy(), bits[:, 1], bits[:, 2] != 0, bits[:, 3].astype(np.int32)

    emit = np.where(((depth == 0) | isspec_in)[:, None], emis, F(0)).astype(F)   # rchit:83
    V = _norm(-D)
    N = Nrm
    ratio = F(0.5) * (F(1) - metal_f)                                            # rchit:127
    rough = np.clip(rough_f, F(0.01), F(0.99))
    metal = np.clip(metal_f, F(0.01), F(0.99))
    seed, r1 = _rnd(seed)
    diffuse = r1 < ratio

    # ---- diffuse lobe (rchit:131-186) ----
    sd, rl = _rnd(seed)
    _ = (rl * lcount.astype(F)).astype(np.int32)   # light index (all candidate lights identical in the harness)
    ldir = lpos - P
    ldist = np.sqrt(_dot(ldir, ldir))
    L = _norm(ldir)
    lit = _dot(L, N) > 0
    # directLight (gltf.glsl:136-154) with the UNCLAMPED material (computePBR_BRDF re-reads it)
    Ld = ldir / ldist[:, None]
    Hh = _norm(Ld + V)
    Li = lcol * lint[:, None] / (ldist * ldist)[:, None]
    cosT = np.maximum(_dot(Ld, N), F(0))
    F0u = F(0.04) * (F(1) - metal_f)[:, None] + base * metal_f[:, None]
    Fr = _schlick(Hh, V, F0u)
    alpha_u = rough_f * rough_f
    k_u = (rough_f + F(1)) * (rough_f + F(1)) / F(8)
    down = F(4) * np.abs(_dot(V, N)) * np.abs(_dot(Ld, N)) + F(1e-4)
    ct = _ndf(N, Hh, alpha_u)[:, None] * Fr * _smith(N, V, Ld, k_u)[:, None] / down[:, None]
    kD = (F(1) - Fr) * (F(1) - metal_f)[:, None]
    brdf_nee = kD * (base * INV_PI) + ct
    brdf_nee = np.where(((ltype == 0) & (cosT > 0))[:, None], brdf_nee, F(0))
    Li = np.where((ltype == 0)[:, None], Li, F(0))
    cosT = np.where(ltype == 0, cosT, F(0))
    nee = lcount.astype(F)[:, None] * brdf_nee * Li * cosT[:, None]
    emit_d = emit + np.where(lit[:, None], nee, F(0))
    sd, h1 = _rnd(sd)
    sd, h2 = _rnd(sd)
    sq = np.sqrt(h1)
    ang = F(2) * PI * h2
    loc = np.stack([np.cos(ang) * sq, np.sin(ang) * sq, np.sqrt(F(1) - h1)], 1).astype(F)
    dir_d = _norm(loc[:, 0:1] * T + loc[:, 1:2] * B + loc[:, 2:3] * N)
    pdf_d = ratio * _dot(dir_d, N) * INV_PI
    brdf_d = (F(1) - metal)[:, None] * base * INV_PI

    # ---- specular lobe (rchit:187-206) ----
    ss, g1 = _rnd(seed)
    ss, g2 = _rnd(ss)
    a2 = (rough * rough) * (rough * rough)
    cth = np.sqrt((F(1) - g2) / ((a2 - F(1)) * g2 + F(1)))
    sth = np.clip(np.sqrt(F(1) - cth * cth), F(0), F(1))
    phi = g1 * F(2) * PI
    hl = np.stack([sth * np.cos(phi), sth * np.sin(phi), cth], 1).astype(F)
    Hs = _norm(hl[:, 0:1] * T + hl[:, 1:2] * B + hl[:, 2:3] * N)
    I = -V
    Ls = _norm(I - (F(2) * _dot(Hs, I))[:, None] * Hs)
    F0 = F(0.04) * (F(1) - metal)[:, None] + base * metal[:, None]
    k = (rough + F(1)) * (rough + F(1)) / F(8)
    pdf_s = (F(1) - ratio) * _dot(N, Hs) / (F(4) * _dot(Ls, Hs) + F(1e-4))
    down_s = F(4) * np.abs(_dot(V, N)) * np.abs(_dot(Ls, N)) + F(1e-4)
    brdf_s = (_schlick(Hs, V, F0) * _smith(N, V, Ls, k)[:, None] / down_s[:, None]) / pdf_s[:, None]

    dsel = diffuse[:, None]
    ray_dir = np.where(dsel, dir_d, Ls)
    brdf = np.where(dsel, brdf_d, brdf_s)
    pdf = np.where(diffuse, pdf_d, F(1))
    cos_out = _dot(ray_dir, N)
    out = np.zeros((n, 20), F)
    out[:, 0:3] = np.where(dsel, emit_d, emit)
    out[:, 3:6] = P
    out[:, 6:9] = ray_dir
    out[:, 9:12] = brdf * cos_out[:, None] / pdf[:, None]
    out[:, 12] = np.where(diffuse, F(0), F(1))
    out[:, 13] = np.where(diffuse, ldist, F(0))
    out[:, 14:17] = np.where(dsel, L, F(0))
    out[:, 17] = np.where(diffuse, sd, ss).astype(np.uint32).view(F)
    return out


def random_records(n, rng):
    """Well-conditioned random shading inputs (orthonormal frames, view above the surface)."""
    def unit(v):
        return v / np.linalg.norm(v, axis=1, keepdims=True)
    N = unit(rng.standard_normal((n, 3)))
    T = unit(np.cross(N, unit(rng.standard_normal((n, 3)))))
    B = np.cross(N, T) * np.where(rng.uniform(size=(n, 1)) < 0.5, -1.0, 1.0)
    Vv = unit(N * rng.uniform(0.2, 1.0, (n, 1)) + 0.8 * unit(rng.standard_normal((n, 3))))
    Vv = np.where(np.sum(Vv * N, 1, keepdims=True) < 0.1, unit(Vv + N), Vv)
    rec = np.zeros((n, 40), F)
    rec[:, 0:3] = rng.uniform(-5, 5, (n, 3))
    rec[:, 3:6], rec[:, 6:9], rec[:, 9:12] = N, T, B
    rec[:, 12:15] = -Vv * rng.uniform(0.5, 2.0, (n, 1))
    rec[:, 15:19] = rng.uniform(0.05, 1.0, (n, 4))
    rec[:, 19] = rng.choice([0.0, 0.0, 0.3, 0.7, 1.0], n)
    rec[:, 20] = rng.uniform(0.0, 1.0, n)
    rec[:, 21:24] = rng.uniform(0, 4, (n, 3)) * (rng.uniform(size=(n, 1)) < 0.3)
    rec[:, 24:27] = rec[:, 0:3] + unit(N + 0.9 * unit(rng.standard_normal((n, 3)))) * rng.uniform(1.0, 9.0, (n, 1))
    rec[:, 27:30] = rng.uniform(0.1, 1.0, (n, 3))
    rec[:, 30] = rng.uniform(10, 100, n)
    ltype = np.where(rng.uniform(size=n) < 0.9, 0, 1).astype(np.int32)
    rec[:, 31] = ltype.view(F)
    bits = np.zeros((n, 4), np.uint32)
    bits[:, 0] = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    bits[:, 1] = rng.integers(0, 4, n)
    bits[:, 2] = rng.integers(0, 2, n)
    bits[:, 3] = rng.choice([1, 1, 8], n)
    rec[:, 32:36] = bits.view(F)
    return rec
