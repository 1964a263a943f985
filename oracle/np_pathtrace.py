"""Second, independent restatement of the CONTROL FLOW of the path (TEST INFRASTRUCTURE).

numpy float32, vectorised over pixels, written from the GLSL -- shaders/raytrace.rgen:24-146 (sample / segment loop,
shadow-ray rule, firefly clamp, stale-payload behaviour on a miss, frame blending), raytrace.rchit:31-113 (attribute
fetch, interpolation, object->world, emissive / normal-map / base-colour / metallic-roughness texture taps),
raytrace.rmiss:11-19, raytraceShadow.rmiss:9-12, raytraceHybrid.rgen:50-286, vert_shader.vert:60-74 +
frag_shader.frag:96-214 -- without consulting oracle.cpp.  The BRDF / sampling half of the hit shader
(raytrace.rchit:115-219, gltf.glsl:55-154, random.glsl:35-70) is np_shading.shade, the first independent restatement.

Ray queries are brute force over all instanced triangles in float64 (closest hit = smallest t in (tmin, tmax), ties ->
smallest flattened triangle id, instance-major; shadow = any hit): no tree, no conservative padding, none of the
product's or the oracle's pre-transformed records.  texture() is the Vulkan-spec bilinear filter with REPEAT
addressing at LOD 0 and the spec's sRGB decode.  Agreement with oracle.cpp / the HIP kernels is therefore to
rounding (different but equivalent operation orders, numpy sin/cos/pow), not bit for bit; tests/golden/
make_np_pathtrace_fixtures.py commits outputs of this file, tests compare the oracle and the GPU with them.
"""
import numpy as np

import np_shading

F = np.float32
U32 = np.uint32


# ---- shaders/random.glsl:6-33 (integer exact) ----------------------------------------------------------------------
def tea(v0, v1):
    v0 = np.asarray(v0, np.uint64) & 0xFFFFFFFF
    v1 = np.broadcast_to(np.asarray(v1, np.uint64) & 0xFFFFFFFF, v0.shape).copy()
    M = np.uint64(0xFFFFFFFF)
    s0 = np.uint64(0)
    for _ in range(16):
        s0 = (s0 + np.uint64(0x9E3779B9)) & M
        v0 = (v0 + ((((v1 << np.uint64(4)) & M) + np.uint64(0xA341316C)) ^ (v1 + s0) ^ ((v1 >> np.uint64(5)) + np.uint64(0xC8013EA4)))) & M
        v1 = (v1 + ((((v0 << np.uint64(4)) & M) + np.uint64(0xAD90777D)) ^ (v0 + s0) ^ ((v0 >> np.uint64(5)) + np.uint64(0x7E95761E)))) & M
    return v0.astype(U32)


def rnd(seed):
    """returns (new seed, float in [0,1))"""
    s = ((seed.astype(np.uint64) * np.uint64(1664525) + np.uint64(1013904223)) & np.uint64(0xFFFFFFFF)).astype(U32)
    return s, (s & U32(0x00FFFFFF)).astype(F) / F(16777216.0)


def _dot(a, b):
    return (a[:, 0] * b[:, 0] + a[:, 1] * b[:, 1]) + a[:, 2] * b[:, 2]


def _normalize(a):
    return (a * (F(1.0) / np.sqrt(_dot(a, a)))[:, None]).astype(F)


def _cross(a, b):
    return np.stack([a[:, 1] * b[:, 2] - a[:, 2] * b[:, 1], a[:, 2] * b[:, 0] - a[:, 0] * b[:, 2], a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]], 1).astype(F)


def _coordinate_system(N):
    """random.glsl:47-54"""
    ax = np.abs(N[:, 0]) > np.abs(N[:, 1])
    z = np.zeros(N.shape[0], F)
    with np.errstate(divide="ignore", invalid="ignore"):
        a = np.stack([N[:, 2], z, -N[:, 0]], 1) / np.sqrt(N[:, 0] * N[:, 0] + N[:, 2] * N[:, 2])[:, None]
        b = np.stack([z, -N[:, 2], N[:, 1]], 1) / np.sqrt(N[:, 1] * N[:, 1] + N[:, 2] * N[:, 2])[:, None]
    Nt = np.where(ax[:, None], a, b).astype(F)
    return Nt, _cross(N, Nt)


def _sampling_hemisphere(seed, x, y, z):
    """random.glsl:35-45"""
    seed, r1 = rnd(seed)
    seed, r2 = rnd(seed)
    sq = np.sqrt(r1)
    ang = F(2) * np_shading.PI * r2
    d = np.stack([np.cos(ang).astype(F) * sq, np.sin(ang).astype(F) * sq, np.sqrt(F(1) - r1)], 1).astype(F)
    return seed, (d[:, 0:1] * x + d[:, 1:2] * y + d[:, 2:3] * z).astype(F)


def _mat4_vec4(M, v):
    """column-major mat4 (16 floats) times (N,4)"""
    M = np.asarray(M, F).reshape(4, 4)  # M[c][r]
    return (((v[:, 0:1] * M[0][None, :] + v[:, 1:2] * M[1][None, :]) + v[:, 2:3] * M[2][None, :]) + v[:, 3:4] * M[3][None, :]).astype(F)


class NpScene:
    """The flat arrays the reference uploads (hello_vulkan.cpp:353-379) + world-space triangles of every TLAS instance
    (hello_vulkan.cpp:1035-1043) for the brute-force ray queries."""

    def __init__(self, flat):
        self.flat = flat
        pos = np.asarray(flat.positions, np.float64).reshape(-1, 3)
        tris, inst_of, prim_of = [], [], []
        self.o2w, self.w2o = [], []
        for i, nd in enumerate(flat.nodes):
            M = np.asarray(nd["worldMatrix"], F).reshape(4, 4).T  # row-major 4x4 from column-major storage
            self.o2w.append(M[:3, :].astype(F))
            self.w2o.append(np.linalg.inv(M[:3, :3].astype(np.float64)).astype(F))
            pm = flat.prim_meshes[int(nd["primMesh"])]
            n = int(pm["indexCount"]) // 3
            idx = np.asarray(flat.indices[int(pm["firstIndex"]): int(pm["firstIndex"]) + 3 * n], np.int64).reshape(n, 3) + int(pm["vertexOffset"])
            P = pos[idx]  # n,3,3 object space
            Pw = P @ M[:3, :3].astype(np.float64).T + M[:3, 3].astype(np.float64)
            tris.append(Pw)
            inst_of.append(np.full(n, i, np.int64))
            prim_of.append(np.arange(n, dtype=np.int64))
        if tris:
            T = np.concatenate(tris)
            self.v0, self.e1, self.e2 = T[:, 0], T[:, 1] - T[:, 0], T[:, 2] - T[:, 0]
            self.inst_of, self.prim_of = np.concatenate(inst_of), np.concatenate(prim_of)
        else:
            self.v0 = self.e1 = self.e2 = np.zeros((0, 3))
            self.inst_of = self.prim_of = np.zeros(0, np.int64)
        self.o2w = np.array(self.o2w, F).reshape(-1, 3, 4)
        self.w2o = np.array(self.w2o, F).reshape(-1, 3, 3)
        self.rays_closest = 0
        self.rays_shadow = 0
        # decode tables: UNORM i/255 and the sRGB EOTF (Vulkan spec 16.x "sRGB EOTF"; hello_vulkan.cpp:417-443 picks the format)
        c = np.arange(256, dtype=np.float64) / 255.0
        self._unorm = c.astype(F)
        self._srgb = np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4).astype(F)
        self._mips = {}

    # -- ray queries (software stand-in for traceRayEXT; what the driver does is not in the reference) --------------------
    def _mt(self, o, d, chunk=96):
        """Moeller-Trumbore of rays (R) against all triangles (T): yields (slice, t[R',T] with inf where no hit, u, v)."""
        for a in range(0, o.shape[0], chunk):
            O, D = o[a:a + chunk].astype(np.float64)[:, None, :], d[a:a + chunk].astype(np.float64)[:, None, :]
            p = np.cross(D, self.e2[None])
            det = np.einsum("tk,rtk->rt", self.e1, p)
            tv = O - self.v0[None]
            U = np.einsum("rtk,rtk->rt", tv, p)
            q = np.cross(tv, self.e1[None])
            V = np.einsum("rtk,rtk->rt", np.broadcast_to(D, q.shape), q)
            Tt = np.einsum("tk,rtk->rt", self.e2, q)
            with np.errstate(divide="ignore", invalid="ignore"):
                inside = np.where(det > 0, (U >= 0) & (V >= 0) & (U + V <= det), (det < 0) & (U <= 0) & (V <= 0) & (U + V >= det))
                t = np.where(inside, Tt / det, np.inf)
                yield slice(a, a + chunk), t, U / det, V / det

    def closest(self, o, d, tmin=0.001, tmax=10000.0):
        n = o.shape[0]
        self.rays_closest += n
        t_out, u_out, v_out, tri = np.full(n, np.inf), np.zeros(n), np.zeros(n), np.full(n, -1, np.int64)
        if self.v0.shape[0] == 0:
            return t_out.astype(F), u_out.astype(F), v_out.astype(F), tri
        for sl, t, u, v in self._mt(o, d):
            t32 = t.astype(F)  # the hit distance a shader sees is a float: compare in float so equal-t ties exist
            ok = (t32 > F(tmin)) & (t32 < F(tmax))
            t32 = np.where(ok, t32, np.inf)
            k = np.argmin(t32, axis=1)  # first occurrence = smallest flattened id among equal t
            r = np.arange(t32.shape[0])
            hit = np.isfinite(t32[r, k])
            tri[sl] = np.where(hit, k, -1)
            t_out[sl], u_out[sl], v_out[sl] = t32[r, k], u[r, k], v[r, k]
        return t_out.astype(F), u_out.astype(F), v_out.astype(F), tri

    def occluded(self, o, d, tmin, tmax):
        n = o.shape[0]
        self.rays_shadow += n
        out = np.zeros(n, bool)
        if self.v0.shape[0] == 0:
            return out
        tmax = np.broadcast_to(np.asarray(tmax, F), (n,))
        for sl, t, _, _ in self._mt(o, d):
            t32 = t.astype(F)
            out[sl] = np.any((t32 > F(tmin)) & (t32 < tmax[sl][:, None]), axis=1)
        return out

    # -- texture(): VkSampler of hello_vulkan.cpp:448-454 (LINEAR, REPEAT) at LOD 0 ------------------------------------
    def _bilinear(self, img, is_srgb, uv):
        """one LINEAR / REPEAT tap per row of uv (N,2) in the RGBA8 image img (h,w,4) -> (N,4) f32"""
        h, w = img.shape[0], img.shape[1]
        x = uv[:, 0] * F(w) - F(0.5)
        y = uv[:, 1] * F(h) - F(0.5)
        fx, fy = np.floor(x), np.floor(y)
        ax, ay = (x - fx).astype(F), (y - fy).astype(F)
        x0 = np.mod(fx.astype(np.int64), w)
        y0 = np.mod(fy.astype(np.int64), h)
        x1, y1 = np.mod(x0 + 1, w), np.mod(y0 + 1, h)
        lut = self._srgb if is_srgb else self._unorm

        def texel(yy, xx):
            p = img[yy, xx]
            return np.concatenate([lut[p[:, :3]], self._unorm[p[:, 3:4]]], 1)

        t00, t10, t01, t11 = texel(y0, x0), texel(y0, x1), texel(y1, x0), texel(y1, x1)
        wx0, wy0 = (F(1) - ax)[:, None], (F(1) - ay)[:, None]
        return ((t00 * wx0 + t10 * ax[:, None]) * wy0 + (t01 * wx0 + t11 * ax[:, None]) * ay[:, None]).astype(F)

    def texture(self, tex_index, uv):
        """tex_index (N,) int (all >= 0 and valid), uv (N,2) f32 -> (N,4) f32"""
        out = np.ones((uv.shape[0], 4), F)
        for ti in np.unique(tex_index):
            m = tex_index == ti
            if ti < 0 or ti >= len(self.flat.textures):
                continue  # the 1x1 white dummy (hello_vulkan.cpp:468-472)
            tx = self.flat.textures[int(ti)]
            out[m] = self._bilinear(np.asarray(tx["rgba8"]), bool(tx["is_srgb"]), uv[m])
        return out

    # -- the fragment shader's texture(): implicit LOD over the mip chain + anisotropy 4 (hello_vulkan.cpp:448-454, :499) --
    def mip_chain(self, ti):
        """Levels of texture ti: level L = linear blit of level L-1 to max(1, size // 2) (nvvk::cmdGenerateMipmaps), float64."""
        if ti in self._mips:
            return self._mips[ti]
        tx = self.flat.textures[int(ti)]
        srgb = bool(tx["is_srgb"])
        chain = [np.asarray(tx["rgba8"])]
        while chain[-1].shape[0] > 1 or chain[-1].shape[1] > 1:
            src = chain[-1]
            sh, sw = src.shape[0], src.shape[1]
            dh, dw = max(1, sh // 2), max(1, sw // 2)
            val = np.empty((sh, sw, 4), np.float64)            # decoded texel values (sRGB images blit in linear space)
            val[..., :3] = (self._srgb if srgb else self._unorm)[src[..., :3]]
            val[..., 3] = self._unorm[src[..., 3]]
            # destination texel centres in unnormalised source coordinates, clamp-to-edge bilinear
            sx = (np.arange(dw) + 0.5) * (sw / dw) - 0.5
            sy = (np.arange(dh) + 0.5) * (sh / dh) - 0.5
            x0f, y0f = np.floor(sx), np.floor(sy)
            wx, wy = sx - x0f, sy - y0f
            xa, xb = np.clip(x0f, 0, sw - 1).astype(np.int64), np.clip(x0f + 1, 0, sw - 1).astype(np.int64)
            ya, yb = np.clip(y0f, 0, sh - 1).astype(np.int64), np.clip(y0f + 1, 0, sh - 1).astype(np.int64)
            top = val[ya][:, xa] * (1 - wx)[None, :, None] + val[ya][:, xb] * wx[None, :, None]
            bot = val[yb][:, xa] * (1 - wx)[None, :, None] + val[yb][:, xb] * wx[None, :, None]
            lin = top * (1 - wy)[:, None, None] + bot * wy[:, None, None]
            if srgb:
                c = lin[..., :3]
                lin[..., :3] = np.where(c <= 0.0031308, 12.92 * c, 1.055 * np.power(np.maximum(c, 0.0), 1.0 / 2.4) - 0.055)
            chain.append(np.floor(np.clip(lin, 0.0, 1.0) * 255.0 + 0.5).astype(np.uint8))
        self._mips[ti] = chain
        return chain

    def texture_grad(self, tex_index, uv, grad):
        """texture() with screen-space derivatives grad (N,4) = (dudx, dvdx, dudy, dvdy).  Vulkan 1.3 spec, Image Operations:
        rho per axis in texels, eta = min(rho_max / rho_min, 4), N = ceil(eta) taps along the major axis at offsets
        i / (N + 1) - 1/2, lambda = log2(rho_max / eta) clamped to the chain, linear blend of the two levels around it."""
        out = np.ones((uv.shape[0], 4), F)
        for ti in np.unique(tex_index):
            m = np.nonzero(tex_index == ti)[0]
            if ti < 0 or ti >= len(self.flat.textures):
                continue
            tx = self.flat.textures[int(ti)]
            srgb = bool(tx["is_srgb"])
            chain = self.mip_chain(int(ti))
            if len(chain) == 1:
                out[m] = self._bilinear(chain[0], srgb, uv[m])
                continue
            h, w = chain[0].shape[0], chain[0].shape[1]
            g = grad[m].astype(np.float64)
            rx = np.hypot(g[:, 0] * w, g[:, 1] * h)
            ry = np.hypot(g[:, 2] * w, g[:, 3] * h)
            major_x = rx >= ry
            rmax, rmin = np.where(major_x, rx, ry), np.where(major_x, ry, rx)
            with np.errstate(divide="ignore", invalid="ignore"):
                eta = np.where(rmin > 0, np.minimum(rmax / rmin, 4.0), np.where(rmax > 0, 4.0, 1.0))
                eta = np.where(eta >= 1.0, eta, 1.0)
                taps = np.ceil(eta).astype(np.int64)
                lam = np.where(rmax / eta > 1.0, np.log2(np.maximum(rmax / eta, 1e-300)), 0.0)
            lam = np.clip(np.nan_to_num(lam, nan=len(chain) - 1.0, posinf=len(chain) - 1.0), 0.0, len(chain) - 1.0)
            hi = np.floor(lam).astype(np.int64)
            lo = np.minimum(hi + 1, len(chain) - 1)
            delta = (lam - hi)[:, None]
            axis = np.where(major_x[:, None], g[:, 0:2], g[:, 2:4])
            acc = np.zeros((m.size, 4), np.float64)
            for i in range(1, 5):
                live = np.nonzero(taps >= i)[0]
                if live.size == 0:
                    break
                off = (i / (taps[live] + 1.0) - 0.5)[:, None]
                tuv = (uv[m][live].astype(np.float64) + off * axis[live]).astype(F)
                a, b = np.zeros((live.size, 4)), np.zeros((live.size, 4))
                for lv in np.unique(np.concatenate([hi[live], lo[live]])):
                    sel = hi[live] == lv
                    if sel.any():
                        a[sel] = self._bilinear(chain[lv], srgb, tuv[sel])
                    sel = lo[live] == lv
                    if sel.any():
                        b[sel] = self._bilinear(chain[lv], srgb, tuv[sel])
                acc[live] += a * (1.0 - delta[live]) + b * delta[live]
            out[m] = (acc / taps[:, None]).astype(F)
        return out

    # -- raytrace.rchit:34-79: attribute fetch and interpolation ----------------------------------------------------------
    def hit_attributes(self, tri, u, v):
        fl = self.flat
        inst, prim = self.inst_of[tri], self.prim_of[tri]
        pmi = np.asarray(fl.nodes["primMesh"], np.int64)[inst]  # gl_InstanceCustomIndexEXT (hello_vulkan.cpp:1038)
        pm = fl.prim_meshes[pmi]
        index_offset = pm["firstIndex"].astype(np.int64) + 3 * prim
        vo = pm["vertexOffset"].astype(np.int64)
        mat_index = np.maximum(0, pm["materialIndex"].astype(np.int64))
        ind = np.asarray(fl.indices, np.int64)
        i0, i1, i2 = ind[index_offset] + vo, ind[index_offset + 1] + vo, ind[index_offset + 2] + vo
        bx, by, bz = (F(1.0) - u - v).astype(F)[:, None], u[:, None], v[:, None]
        P, N_, T_, UV = (np.asarray(a, F) for a in (fl.positions, fl.normals, fl.tangents, fl.texcoords0))

        def interp(A):
            return (A[i0] * bx + A[i1] * by + A[i2] * bz).astype(F)

        pos = interp(P.reshape(-1, 3))
        o2w, w2o = self.o2w[inst], self.w2o[inst]
        world_pos = (((o2w[:, :, 0] * pos[:, 0:1] + o2w[:, :, 1] * pos[:, 1:2]) + o2w[:, :, 2] * pos[:, 2:3]) + o2w[:, :, 3]).astype(F)

        def n_times_w2o(n):  # vec3(n * gl_WorldToObjectEXT): component j = dot(n, column j)
            return ((n[:, 0:1] * w2o[:, 0, :] + n[:, 1:2] * w2o[:, 1, :]) + n[:, 2:3] * w2o[:, 2, :]).astype(F)

        nrm = _normalize(interp(N_.reshape(-1, 3)))
        world_nrm = _normalize(n_times_w2o(nrm))
        T4 = T_.reshape(-1, 4)
        tag = _normalize(interp(T4[:, :3]))
        world_tag = _normalize(n_times_w2o(tag))
        world_tag = _normalize(world_tag - _dot(world_tag, world_nrm)[:, None] * world_nrm)
        world_bin = (T4[i0, 3][:, None] * _cross(world_nrm, world_tag)).astype(F)
        tex_coord = interp(UV.reshape(-1, 2))
        return dict(world_pos=world_pos, world_nrm=world_nrm, world_tag=world_tag, world_bin=world_bin, uv=tex_coord,
                    mat=fl.materials[mat_index], inst=inst, i=(i0, i1, i2), b=(bx, by, bz))

    def material_inputs(self, mat, uv, grad=None):
        """pbrGetBaseColor / pbrGetMetallicRoughness (gltf.glsl:26-45); grad: fragment-shader derivatives (G-buffer) or None"""
        tex = (lambda ti, tuv, sel: self.texture(ti, tuv)) if grad is None else (lambda ti, tuv, sel: self.texture_grad(ti, tuv, grad[sel]))
        base = np.asarray(mat["pbrBaseColorFactor"], F)[:, :3].copy()
        bt = mat["pbrBaseColorTexture"].astype(np.int64)
        m = bt > -1
        if m.any():
            base[m] = base[m] * tex(bt[m], uv[m], m)[:, :3]
        metal, rough = mat["metallicFactor"].astype(F).copy(), mat["roughnessFactor"].astype(F).copy()
        mt = mat["metallicRoughnessTexture"].astype(np.int64)
        m = mt > -1
        if m.any():
            t = tex(mt[m], uv[m], m)
            rough[m] = rough[m] * t[:, 1]
            metal[m] = metal[m] * t[:, 2]
        return base.astype(F), metal, rough


class _Payload:
    def __init__(self, n):
        self.hitValue = np.zeros((n, 3), F)
        self.seed = np.zeros(n, U32)
        self.depth = np.zeros(n, np.int64)
        self.rayOrigin = np.zeros((n, 3), F)
        self.rayDirection = np.zeros((n, 3), F)
        self.weight = np.zeros((n, 3), F)
        self.isSpecular = np.zeros(n, bool)  # (uninitialised in the GLSL before the first hit; never read before it is written)
        self.lightDist = np.zeros(n, F)
        self.shadowRayDir = np.zeros((n, 3), F)


def _closest_hit(sc, pc, prd, lanes, tri, u, v, ray_dir):
    """raytrace.rchit:31-219 for the lanes `lanes` (indices into prd) that hit triangle `tri`."""
    A = sc.hit_attributes(tri, u, v)
    mat, uv = A["mat"], A["uv"]
    n = lanes.shape[0]
    emits = (prd.depth[lanes] == 0) | prd.isSpecular[lanes]                      # rchit:83
    emittance = np.where(emits[:, None], np.asarray(mat["emissiveFactor"], F), F(0)).astype(F)
    et = mat["emissiveTexture"].astype(np.int64)
    m = emits & (et > -1)
    if m.any():
        emittance[m] = emittance[m] * sc.texture(et[m], uv[m])[:, :3]
    tangent, binormal, tex_normal = A["world_tag"].copy(), A["world_bin"].copy(), A["world_nrm"].copy()
    nt = mat["normalTexture"].astype(np.int64)
    m = nt > -1
    if m.any():                                                                   # rchit:100-106
        tn = _normalize(sc.texture(nt[m], uv[m])[:, :3] * F(2.0) - F(1.0))
        tn = _normalize(tangent[m] * tn[:, 0:1] + binormal[m] * tn[:, 1:2] + tex_normal[m] * tn[:, 2:3])
        tex_normal[m] = tn
        t2, b2 = _coordinate_system(tn)
        tangent[m], binormal[m] = t2, b2
    base, metal, rough = sc.material_inputs(mat, uv)
    # light of the diffuse lobe: second rnd of the hit (rchit:130,137); picked here so np_shading sees the light it needs
    s1, _ = rnd(prd.seed[lanes])
    _, rl = rnd(s1)
    li = (rl * F(pc.lightsCount)).astype(np.int64)
    L = sc.flat.lights[np.clip(li, 0, len(sc.flat.lights) - 1)]
    rec = np.zeros((n, 40), F)
    rec[:, 0:3], rec[:, 3:6], rec[:, 6:9], rec[:, 9:12], rec[:, 12:15] = A["world_pos"], tex_normal, tangent, binormal, ray_dir
    rec[:, 15:18], rec[:, 19], rec[:, 20], rec[:, 21:24] = base, metal, rough, emittance
    rec[:, 24:27], rec[:, 27:30], rec[:, 30] = np.asarray(L["position"], F), np.asarray(L["color"], F), L["intensity"].astype(F)
    rec[:, 31] = L["type"].astype(np.int32).view(F)
    bits = np.zeros((n, 4), U32)
    bits[:, 0], bits[:, 1], bits[:, 2], bits[:, 3] = prd.seed[lanes], 1, 1, pc.lightsCount  # depth / isSpecular: emission already resolved above
    rec[:, 32:36] = bits.view(F)
    out = np_shading.shade(rec)
    prd.hitValue[lanes] = out[:, 0:3]
    prd.rayOrigin[lanes] = out[:, 3:6]
    prd.rayDirection[lanes] = out[:, 6:9]
    prd.weight[lanes] = out[:, 9:12]
    spec = out[:, 12] != 0
    prd.isSpecular[lanes] = spec
    d = ~spec
    ld, sd = prd.lightDist[lanes], prd.shadowRayDir[lanes]
    ld[d], sd[d] = out[d, 13], out[d, 14:17]  # only the diffuse branch writes them (rchit:143-144)
    prd.lightDist[lanes], prd.shadowRayDir[lanes] = ld, sd
    prd.seed[lanes] = out[:, 17].copy().view(U32)
    return A


def _segment_loop(sc, pc, prd, cur_weight, hit_value, clear_color):
    """raytrace.rgen:62-116 == raytraceHybrid.rgen:206-265: `for (; prd.depth < pcRay.depth; prd.depth++)`."""
    while True:
        act = np.nonzero(prd.depth < pc.depth)[0]
        if act.size == 0:
            break
        t, u, v, tri = sc.closest(prd.rayOrigin[act], prd.rayDirection[act])
        hit = tri >= 0
        if hit.any():
            _closest_hit(sc, pc, prd, act[hit], tri[hit], u[hit], v[hit], prd.rayDirection[act[hit]])
        ms = act[~hit]
        if ms.size:                                                               # raytrace.rmiss:13-17
            first = prd.depth[ms] == 0
            prd.hitValue[ms] = np.where(first[:, None], (np.asarray(clear_color[:3], F) * F(0.8))[None, :], F(0.01)).astype(F)
            prd.depth[ms] = 100
        shadow_hit = np.zeros(act.size, bool)
        sh = ~prd.isSpecular[act] & (prd.depth[act] != 100)                      # rgen:79
        if sh.any():
            ls = act[sh]
            shadow_hit[sh] = sc.occluded(prd.rayOrigin[ls], prd.shadowRayDir[ls], 0.001, prd.lightDist[ls] - F(0.1))
        lit = act[~shadow_hit]
        hit_value[lit] = hit_value[lit] + np.minimum(prd.hitValue[lit] * cur_weight[lit], F(10.0))  # rgen:99-102
        cur_weight[act] = cur_weight[act] * prd.weight[act]                      # rgen:115 (stale weight after a miss included)
        prd.depth[act] += 1


def pathtrace_pixels(sc, pc, view_inverse, proj_inverse, W, H, seed, xs, ys, old=None, row_major_seed=False):
    """raytrace.rgen:24-146 for the pixels (xs, ys) of a W x H launch; clockARB() -> `seed`.  Returns rgba (N,4) f32."""
    xs, ys = np.asarray(xs, np.int64), np.asarray(ys, np.int64)
    n = xs.shape[0]
    prd = _Payload(n)
    index = (ys * W + xs) if row_major_seed else (ys * xs + xs)
    prd.seed = tea(index & 0xFFFFFFFF, seed)
    hit_values = np.zeros((n, 3), F)
    origin = _mat4_vec4(view_inverse, np.tile(np.array([[0, 0, 0, 1]], F), (n, 1)))
    clear = [pc.clearColor[k] for k in range(4)]
    for _ in range(pc.samples):
        prd.seed, r1 = rnd(prd.seed)
        prd.seed, r2 = rnd(prd.seed)
        jx, jy = (np.full(n, F(0.5)), np.full(n, F(0.5))) if pc.frame == 0 else (r1, r2)
        in_u = ((xs.astype(F) + jx) / F(W)).astype(F)
        in_v = ((ys.astype(F) + jy) / F(H)).astype(F)
        dx, dy = in_u * F(2.0) - F(1.0), in_v * F(2.0) - F(1.0)
        target = _mat4_vec4(proj_inverse, np.stack([dx, dy, np.ones(n, F), np.ones(n, F)], 1).astype(F))
        tn = _normalize(target[:, :3])
        direction = _mat4_vec4(view_inverse, np.concatenate([tn, np.zeros((n, 1), F)], 1))
        prd.hitValue[:] = 0
        prd.rayOrigin[:] = origin[:, :3]
        prd.rayDirection[:] = direction[:, :3]
        prd.depth[:] = 0
        prd.weight[:] = 0
        cur_weight = np.ones((n, 3), F)
        hit_value = np.zeros((n, 3), F)
        _segment_loop(sc, pc, prd, cur_weight, hit_value, clear)
        hit_values = hit_values + hit_value
    res = (hit_values / F(pc.samples)).astype(F)
    out = np.ones((n, 4), F)
    if pc.frame > 0:
        a = F(1.0) / F(pc.frame + 1)
        out[:, :3] = old[:, :3] * (F(1.0) - a) + res * a  # mix()
    else:
        out[:, :3] = res
    return out


# ---- hybrid mode -----------------------------------------------------------------------------------------------------
def _half(x):
    return np.asarray(x, F).astype(np.float16).astype(F)  # rg16f render target (hello_vulkan.cpp:650-741), RNE


def _pixel_directions(view_inverse, proj_inverse, W, H, xs, ys):
    n = xs.shape[0]
    in_u = ((xs.astype(F) + F(0.5)) / F(W)).astype(F)
    in_v = ((ys.astype(F) + F(0.5)) / F(H)).astype(F)
    target = _mat4_vec4(proj_inverse, np.stack([in_u * F(2) - F(1), in_v * F(2) - F(1), np.ones(n, F), np.ones(n, F)], 1).astype(F))
    return _mat4_vec4(view_inverse, np.concatenate([_normalize(target[:, :3]), np.zeros((n, 1), F)], 1))[:, :3]


def _quad_derivatives(origin, corners, corner_uv, uv, view_inverse, proj_inverse, W, H, xs, ys):
    """dFdx / dFdy of fragTexCoord: the other fragment of the pixel's 2x2 quad in x (y), run on the same primitive (a helper
    invocation extrapolates the primitive's plane), minus this one, signed so the difference points to +x (+y).  float64."""
    p0, p1, p2 = (c.astype(np.float64) for c in corners)
    e1, e2 = p1 - p0, p2 - p0
    nrm = np.cross(e1, e2)
    o = origin.astype(np.float64)
    out = np.zeros((xs.shape[0], 4), np.float64)
    for k, (nx, ny, odd) in enumerate(((xs ^ 1, ys, xs & 1), (xs, ys ^ 1, ys & 1))):
        d = _pixel_directions(view_inverse, proj_inverse, W, H, nx, ny).astype(np.float64)
        denom = np.einsum("ij,ij->i", nrm, d)
        ok = denom != 0
        t = np.einsum("ij,ij->i", nrm, p0 - o) / np.where(ok, denom, 1.0)
        q = o + d * t[:, None] - p0                            # point on the plane relative to p0; barycentrics by Cramer
        a11, a12, a22 = (e1 * e1).sum(1), (e1 * e2).sum(1), (e2 * e2).sum(1)
        b1, b2 = (q * e1).sum(1), (q * e2).sum(1)
        det = a11 * a22 - a12 * a12
        ok &= det != 0
        det = np.where(ok, det, 1.0)
        bu, bv = (b1 * a22 - b2 * a12) / det, (b2 * a11 - b1 * a12) / det
        nuv = corner_uv[0] * (1 - bu - bv)[:, None] + corner_uv[1] * bu[:, None] + corner_uv[2] * bv[:, None]
        diff = (nuv - uv.astype(np.float64)) * np.where(odd == 1, -1.0, 1.0)[:, None]
        out[:, 2 * k: 2 * k + 2] = np.where(ok[:, None], diff, 0.0)
    return out


def gbuffer_pixels(sc, clear_color, lights_count, view_inverse, proj_inverse, W, H, xs, ys, mips=True):
    """The raster pass as a primary-ray cast through pixel centres: vert_shader.vert:60-74 per vertex, barycentric
    interpolation, frag_shader.frag:96-214.  mips: texture() takes its implicit LOD from the quad derivatives of fragTexCoord
    (a fragment shader; False = LOD 0).  Returns dict of planes (N,4)/(N,2)."""
    xs, ys = np.asarray(xs, np.int64), np.asarray(ys, np.int64)
    n = xs.shape[0]
    color = np.tile(np.asarray(clear_color, F)[None, :], (n, 1))                  # main.cpp:483-487 clear values
    position = np.tile(np.array([[0, 0, 0, 1]], F), (n, 1))
    normal = np.tile(np.array([[0, 0, 0, 1]], F), (n, 1))
    rough = np.zeros((n, 2), F)
    origin = _mat4_vec4(view_inverse, np.tile(np.array([[0, 0, 0, 1]], F), (n, 1)))[:, :3]
    direction = _pixel_directions(view_inverse, proj_inverse, W, H, xs, ys)
    t, u, v, tri = sc.closest(origin, direction)
    h = np.nonzero(tri >= 0)[0]
    if h.size == 0:
        return dict(color=color, position=position, normal=normal, roughMetal=rough)
    A = sc.hit_attributes(tri[h], u[h], v[h])
    fl = sc.flat
    P, N_, T4 = np.asarray(fl.positions, F).reshape(-1, 3), np.asarray(fl.normals, F).reshape(-1, 3), np.asarray(fl.tangents, F).reshape(-1, 4)
    o2w, w2o = sc.o2w[A["inst"]], sc.w2o[A["inst"]]
    w_pos = np.zeros((h.size, 3), F); w_nrm = np.zeros((h.size, 3), F); w_tag = np.zeros((h.size, 3), F); w_bin = np.zeros((h.size, 3), F)
    corners, corner_uv = [], []
    UV = np.asarray(fl.texcoords0, F).reshape(-1, 2)
    for i, b in zip(A["i"], A["b"]):                                              # vert_shader.vert:62-72 at each of the three vertices
        p = P[i]
        wp = (((o2w[:, :, 0] * p[:, 0:1] + o2w[:, :, 1] * p[:, 1:2]) + o2w[:, :, 2] * p[:, 2:3]) + o2w[:, :, 3]).astype(F)
        corners.append(wp)
        corner_uv.append(UV[i].astype(np.float64))

        def it(nv):  # mat3(inverseTransposeMatrix) * n
            return ((nv[:, 0:1] * w2o[:, 0, :] + nv[:, 1:2] * w2o[:, 1, :]) + nv[:, 2:3] * w2o[:, 2, :]).astype(F)

        wn = _normalize(it(N_[i]))
        wt = _normalize(it(T4[i, :3]))
        wt = _normalize(wt - _dot(wt, wn)[:, None] * wn)
        wb = (_cross(wn, wt) * T4[i, 3][:, None]).astype(F)
        w_pos, w_nrm, w_tag, w_bin = w_pos + wp * b, w_nrm + wn * b, w_tag + wt * b, w_bin + wb * b
    mat, uv = A["mat"], A["uv"]
    grad = _quad_derivatives(origin[h], corners, corner_uv, uv, view_inverse, proj_inverse, W, H, xs[h], ys[h]) if mips else None
    tex = (lambda ti, sel: sc.texture(ti, uv[sel])) if grad is None else (lambda ti, sel: sc.texture_grad(ti, uv[sel], grad[sel]))
    view_dir = (w_pos - origin[h]).astype(F)
    N = _normalize(w_nrm)                                                         # frag_shader.frag:96-119
    nt = mat["normalTexture"].astype(np.int64)
    m = nt > -1
    if m.any():
        T = _normalize(w_tag[m]); B = _normalize(w_bin[m]); Nm = N[m]
        T = _normalize(T - _dot(T, Nm)[:, None] * Nm)
        B = _normalize(B - _dot(B, Nm)[:, None] * Nm - _dot(B, T)[:, None] * T)
        nrm = _normalize(tex(nt[m], m)[:, :3] * F(2.0) - F(1.0))
        N[m] = _normalize(T * nrm[:, 0:1] + B * nrm[:, 1:2] + Nm * nrm[:, 2:3])
    base, metal, rough_v = sc.material_inputs(mat, uv, grad)
    albedo = ((F(1.0) - metal)[:, None] * base).astype(F)
    V = _normalize(-view_dir)
    emit = np.asarray(mat["emissiveFactor"], F).copy()
    et = mat["emissiveTexture"].astype(np.int64)
    m = et > -1
    if m.any():
        emit[m] = emit[m] * tex(et[m], m)[:, :3]
    col = np.zeros((h.size, 3), F)
    for li in range(lights_count):                                               # frag_shader.frag:193-213
        lt = fl.lights[li]
        lp = np.tile(np.asarray(lt["position"], F)[None, :], (h.size, 1))
        L = _normalize(lp - w_pos)
        inten = np.tile((np.asarray(lt["color"], F) * F(lt["intensity"]))[None, :], (h.size, 1)).astype(F)
        if int(lt["type"]) == 0:
            ld = lp - w_pos
            dd = np.sqrt(_dot(ld, ld))
            inten = (inten / (dd * dd)[:, None]).astype(F)
        else:
            L = _normalize(lp)
        Hh = _normalize(L + V)
        cos_t = np.maximum(_dot(L, N), F(0.0))
        brdf = _pbr_brdf(N, V, L, Hh, base, metal, rough_v)
        col = np.where((cos_t > 0)[:, None], col + brdf * inten * cos_t[:, None], col).astype(F)
    color[h, :3], color[h, 3] = emit + col, albedo[:, 0]
    position[h, :3], position[h, 3] = w_pos, albedo[:, 1]
    normal[h, :3], normal[h, 3] = N, albedo[:, 2]
    rough[h, 0], rough[h, 1] = _half(rough_v), _half(metal)
    return dict(color=color, position=position, normal=normal, roughMetal=rough)


def _pbr_brdf(N, V, L, H, base, metal, rough):
    """computePBR_BRDF, gltf.glsl:111-134, with the UNCLAMPED material (it re-reads the material itself)"""
    one = F(1.0)
    F0 = (F(0.04) * (one - metal)[:, None] + base * metal[:, None]).astype(F)
    Fr = (F0 + (one - F0) * np.power(one - np.abs(_dot(H, V)), F(5.0))[:, None]).astype(F)
    alpha = rough * rough
    k = (rough + one) * (rough + one) / F(8.0)
    a2 = alpha * alpha
    NH = _dot(N, H)
    dd = NH * NH * (a2 - one) + one
    D = np.where(NH <= 0, F(0), a2 * np_shading.INV_PI / (dd * dd + F(1e-4)))
    nv, nl = np.abs(_dot(N, V)), np.abs(_dot(N, L))
    G = (nv / (nv * (one - k) + k)) * (nl / (nl * (one - k) + k))
    down = F(4.0) * np.abs(_dot(V, N)) * np.abs(_dot(L, N)) + F(1e-4)
    ct = D[:, None] * Fr * G[:, None] / down[:, None]
    kD = (one - Fr) * (one - metal)[:, None]
    return (kD * (base * np_shading.INV_PI) + ct).astype(F)


def hybrid_pixels(sc, pc, view_inverse, W, H, seed, xs, ys, g, accum_old=None, row_major_seed=False):
    """raytraceHybrid.rgen:50-286 for pixels (xs, ys) with their G-buffer texels g[plane] (N, ...).  Returns the new imageAccum texels."""
    xs, ys = np.asarray(xs, np.int64), np.asarray(ys, np.int64)
    n = xs.shape[0]
    prd = _Payload(n)
    prd.seed = tea(((ys * W + xs) if row_major_seed else (ys * xs + xs)) & 0xFFFFFFFF, seed)
    color = np.tile(np.array([[0, 0, 0, 1]], F), (n, 1))
    world_pos, world_nrm = np.asarray(g["position"], F)[:, :3].copy(), np.asarray(g["normal"], F)[:, :3].copy()
    shaded = ~(np.all(world_pos == 0, 1) & np.all(world_nrm == 0, 1))           # rgen:67-71
    albedo = np.stack([g["color"][:, 3], g["position"][:, 3], g["normal"][:, 3]], 1).astype(F)
    roughness, metalness = np.asarray(g["roughMetal"], F)[:, 0], np.asarray(g["roughMetal"], F)[:, 1]
    S = np.nonzero(shaded)[0]
    if pc.useShadows == 1 and S.size:                                            # rgen:81-131
        sd, r = rnd(prd.seed[S])
        prd.seed[S] = sd
        li = np.clip((r * F(pc.lightsCount)).astype(np.int64), 0, len(sc.flat.lights) - 1)
        lp = np.asarray(sc.flat.lights["position"], F)[li]
        ldir = (lp - world_pos[S]).astype(F)
        dist = np.sqrt(_dot(ldir, ldir))
        L = _normalize(ldir)
        vis = np.ones(S.size, F)
        back = _dot(L, world_nrm[S]) < 0
        vis[back] = 0
        f = ~back
        if f.any():
            occ = sc.occluded(world_pos[S][f], L[f], 0.1, dist[f] - F(0.1))
            vf = vis[f]; vf[occ] = 0; vis[f] = vf
        color[S, 3] = color[S, 3] * np.maximum(vis, F(0.01))
    if pc.useAO == 1 and S.size:                                                 # rgen:134-169
        ao = np.zeros(S.size, F)
        tg, bn = _coordinate_system(world_nrm[S])
        for _ in range(4):
            sd, d = _sampling_hemisphere(prd.seed[S], tg, bn, world_nrm[S])
            prd.seed[S] = sd
            occ = sc.occluded(world_pos[S], _normalize(d), 0.1, 2.0)
            ao = ao + np.where(occ, F(0.25), F(0)).astype(F)
        color[S, 3] = color[S, 3] * (F(1.0) - ao)
    if pc.useGI == 1 and S.size:                                                 # rgen:172-271
        ratio = metalness[S] * (F(1.0) - roughness[S])
        diff = ratio < F(0.8)
        direction = np.zeros((S.size, 3), F)
        cur_weight_s = np.ones((S.size, 3), F)
        spec_flag = ~diff
        if diff.any():
            D = S[diff]
            tg, bn = _coordinate_system(world_nrm[D])
            sd, d = _sampling_hemisphere(prd.seed[D], tg, bn, world_nrm[D])
            prd.seed[D] = sd
            direction[diff] = _normalize(d)
            cur_weight_s[diff] = albedo[D]
        if spec_flag.any():
            Q = S[spec_flag]
            cam = _mat4_vec4(view_inverse, np.tile(np.array([[0, 0, 0, 1]], F), (Q.size, 1)))[:, :3]
            V = _normalize(cam - world_pos[Q])
            I = -V
            Nn = world_nrm[Q]
            direction[spec_flag] = _normalize(I - (F(2.0) * _dot(Nn, I))[:, None] * Nn)
        prd.isSpecular[S] = spec_flag
        prd.hitValue[S] = 0
        prd.rayOrigin[S] = world_pos[S]
        prd.rayDirection[S] = direction
        prd.depth[:] = 10 ** 6
        prd.depth[S] = 1
        prd.weight[S] = 0
        cur_weight = np.ones((n, 3), F)
        cur_weight[S] = cur_weight_s
        hit_value = np.zeros((n, 3), F)
        _segment_loop(sc, pc, prd, cur_weight, hit_value, [pc.clearColor[k] for k in range(4)])
        color[S, :3] = hit_value[S]
    if pc.frame > 0:                                                             # accumulateFrames, rgen:36-48
        a = F(1.0) / F(pc.frame + 1)
        return (np.asarray(accum_old, F) * (F(1.0) - a) + color * a).astype(F)
    return color
