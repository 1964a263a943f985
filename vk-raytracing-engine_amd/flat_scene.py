"""Flat scene arrays (the data contract the reference uploads, hello_vulkan.cpp:353-379)
as numpy arrays, with .npz io and marshalling into `vkrt_scene_desc` for the C ABI.

Harness-side container only: it holds what `nvh::GltfScene` would hand the reference
(m_positions, m_indices, m_normals, m_tangents, m_texcoords0, m_primMeshes, m_nodes,
m_materials, m_lights) plus decoded RGBA8 textures.
"""
import ctypes as C
from dataclasses import dataclass, field
from typing import List

import numpy as np

from . import abi

MAT_DTYPE = np.dtype(
    [
        ("pbrBaseColorFactor", "<f4", 4),
        ("pbrBaseColorTexture", "<i4"),
        ("metallicFactor", "<f4"),
        ("roughnessFactor", "<f4"),
        ("metallicRoughnessTexture", "<i4"),
        ("normalTexture", "<i4"),
        ("emissiveFactor", "<f4", 3),
        ("emissiveTexture", "<i4"),
    ]
)
LIGHT_DTYPE = np.dtype([("position", "<f4", 3), ("color", "<f4", 3), ("intensity", "<f4"), ("type", "<i4")])
PRIM_DTYPE = np.dtype(
    [("firstIndex", "<u4"), ("indexCount", "<u4"), ("vertexOffset", "<u4"), ("vertexCount", "<u4"), ("materialIndex", "<i4")]
)
NODE_DTYPE = np.dtype([("worldMatrix", "<f4", 16), ("primMesh", "<i4")])
assert MAT_DTYPE.itemsize == 52 and LIGHT_DTYPE.itemsize == 32 and PRIM_DTYPE.itemsize == 20 and NODE_DTYPE.itemsize == 68


def fallback_lights():
    """The 8 hard-coded point lights used when a file has none (hello_vulkan.cpp:247-321)."""
    L = np.zeros(8, LIGHT_DTYPE)
    pos = [(1.0, 5.0, -1.33), (0, 3, 67), (-1.3, 7.62, 59), (2.4, 2.05, 40.6), (-0.33, 6.85, 30),
           (-6.2, 9.6, 20.18), (-0.23, 6.93, 12.21), (0.24, 3.03, 49.94)]
    col = [(1, 1, 1), (1.0, 0.01, 0.1), (1, 1, 1), (1, 1, 1), (1, 1, 1), (1, 1, 1), (1.0, 1.0, 0.0), (0.0, 0.0, 1.0)]
    L["position"] = np.array(pos, np.float32)
    L["color"] = np.array(col, np.float32)
    L["intensity"] = 50.0
    L["type"] = 0
    return L


@dataclass
class FlatScene:
    positions: np.ndarray  # (V,3) f32
    normals: np.ndarray  # (V,3) f32
    tangents: np.ndarray  # (V,4) f32
    texcoords0: np.ndarray  # (V,2) f32
    indices: np.ndarray  # (I,) u32
    prim_meshes: np.ndarray  # PRIM_DTYPE
    materials: np.ndarray  # MAT_DTYPE
    lights: np.ndarray  # LIGHT_DTYPE
    nodes: np.ndarray  # NODE_DTYPE
    textures: List[dict] = field(default_factory=list)  # {"rgba8": (H,W,4) u8, "is_srgb": bool}

    def __post_init__(self):
        self.positions = np.ascontiguousarray(self.positions, np.float32).reshape(-1, 3)
        self.normals = np.ascontiguousarray(self.normals, np.float32).reshape(-1, 3)
        self.tangents = np.ascontiguousarray(self.tangents, np.float32).reshape(-1, 4)
        self.texcoords0 = np.ascontiguousarray(self.texcoords0, np.float32).reshape(-1, 2)
        self.indices = np.ascontiguousarray(self.indices, np.uint32).reshape(-1)
        self.prim_meshes = np.ascontiguousarray(self.prim_meshes, PRIM_DTYPE)
        self.materials = np.ascontiguousarray(self.materials, MAT_DTYPE)
        self.lights = np.ascontiguousarray(self.lights, LIGHT_DTYPE)
        self.nodes = np.ascontiguousarray(self.nodes, NODE_DTYPE)

    # -- stats ---------------------------------------------------------------------------
    @property
    def instanced_triangle_count(self):
        return int(sum(int(self.prim_meshes["indexCount"][n["primMesh"]]) // 3 for n in self.nodes))

    # -- io ------------------------------------------------------------------------------
    def save_npz(self, path):
        d = dict(
            positions=self.positions, normals=self.normals, tangents=self.tangents, texcoords0=self.texcoords0,
            indices=self.indices, prim_meshes=self.prim_meshes, materials=self.materials, lights=self.lights,
            nodes=self.nodes, texture_count=np.int32(len(self.textures)),
        )
        for i, t in enumerate(self.textures):
            d[f"tex{i}_rgba8"] = np.ascontiguousarray(t["rgba8"], np.uint8)
            d[f"tex{i}_srgb"] = np.int32(1 if t["is_srgb"] else 0)
        np.savez_compressed(path, **d)

    @staticmethod
    def load_npz(path):
        z = np.load(path)
        tex = []
        for i in range(int(z["texture_count"])):
            tex.append({"rgba8": z[f"tex{i}_rgba8"], "is_srgb": bool(z[f"tex{i}_srgb"])})
        return FlatScene(z["positions"], z["normals"], z["tangents"], z["texcoords0"], z["indices"],
                         z["prim_meshes"], z["materials"], z["lights"], z["nodes"], tex)

    # -- C ABI marshalling -----------------------------------------------------------------
    def to_desc(self):
        """Returns (vkrt_scene_desc, keepalive).  Pointers stay valid while keepalive lives."""
        keep = [self.positions, self.normals, self.tangents, self.texcoords0, self.indices,
                self.prim_meshes, self.materials, self.lights, self.nodes]
        d = abi.SceneDesc()
        d.struct_size = C.sizeof(abi.SceneDesc)
        d.vertex_count = self.positions.shape[0]
        d.positions = self.positions.ctypes.data
        d.normals = self.normals.ctypes.data
        d.tangents = self.tangents.ctypes.data
        d.texcoords0 = self.texcoords0.ctypes.data
        d.indices = self.indices.ctypes.data
        d.index_count = self.indices.shape[0]
        d.prim_mesh_count = self.prim_meshes.shape[0]
        d.prim_meshes = self.prim_meshes.ctypes.data
        d.materials = self.materials.ctypes.data
        d.material_count = self.materials.shape[0]
        d.light_count = self.lights.shape[0]
        d.lights = self.lights.ctypes.data
        d.nodes = self.nodes.ctypes.data
        d.node_count = self.nodes.shape[0]
        d.texture_count = len(self.textures)
        if self.textures:
            arr = (abi.Texture * len(self.textures))()
            for i, t in enumerate(self.textures):
                px = np.ascontiguousarray(t["rgba8"], np.uint8)
                keep.append(px)
                arr[i].width = px.shape[1]
                arr[i].height = px.shape[0]
                arr[i].rgba8 = px.ctypes.data
                arr[i].is_srgb = 1 if t["is_srgb"] else 0
            keep.append(arr)
            d.textures = C.cast(arr, C.c_void_p)
        else:
            d.textures = None
        return d, keep


def make_push_constants(samples=1, depth=3, frame=0, lights_count=1, clear_color=(1.0, 1.0, 1.0, 1.0)):
    """PushConstantRay with the reference defaults (hello_vulkan.cpp:911-918, main.cpp:247)."""
    pc = abi.PushConstantRay()
    for i in range(4):
        pc.clearColor[i] = clear_color[i]
    pc.frame = frame
    pc.lightsCount = lights_count
    pc.samples = samples
    pc.depth = depth
    pc.useShadows = 1
    pc.useAO = 1
    pc.useGI = 0
    return pc


def uniforms_from_matrices(view_proj, view_inverse, proj_inverse):
    """Column-major GlobalUniforms from three 4x4 numpy matrices given in math (row, col) form."""
    u = abi.GlobalUniforms()
    for name, M in (("viewProj", view_proj), ("viewInverse", view_inverse), ("projInverse", proj_inverse)):
        flat = np.asarray(M, np.float32).T.reshape(-1)  # column-major storage
        dst = getattr(u, name)
        for i in range(16):
            dst.m[i] = float(flat[i])
    return u
