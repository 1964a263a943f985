// gltf_scene.h -- flat glTF scene container of the host layer.
//
// Holds what the reference's loadGltfScene keeps in `nvh::GltfScene m_gltfScene` and uploads
// (reference hello_vulkan.cpp:344-368): m_positions / m_indices / m_normals / m_tangents /
// m_texcoords0, m_primMeshes, m_nodes, m_materials, m_lights -- plus decoded RGBA8 textures
// (createTextureImages, hello_vulkan.cpp:445-513).  Produced by gltf_loader.cpp.
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include "../../include/vkrt.h"

namespace vkrt_host {

struct TextureImage
{
  uint32_t width = 1, height = 1;
  std::vector<uint8_t> rgba;  // width*height*4
  bool srgb = false;
};

struct GltfScene
{
  std::vector<float> m_positions;   // vec3
  std::vector<float> m_normals;     // vec3
  std::vector<float> m_tangents;    // vec4
  std::vector<float> m_texcoords0;  // vec2
  std::vector<uint32_t> m_indices;
  std::vector<vkrt_prim_mesh> m_primMeshes;
  std::vector<vkrt_node> m_nodes;
  std::vector<GltfPBRMaterial> m_materials;  // already in shader layout (loadGltfMaterials :207-224)
  std::vector<GltfLight> m_lights;           // incl. the 8 fallback lights (loadGltfLights :226-325)
  std::vector<TextureImage> m_textures;      // one per glTF texture (:505-509)
  std::string warnings;

  uint32_t vertexCount() const { return (uint32_t)(m_positions.size() / 3); }
  uint32_t instancedTriangleCount() const
  {
    uint32_t n = 0;
    for(const auto& nd : m_nodes) n += m_primMeshes[nd.primMesh].indexCount / 3;
    return n;
  }
  // Borrowing view for vkrt_scene_create; `tex` receives the texture table it points into.
  vkrt_scene_desc desc(std::vector<vkrt_texture>& tex) const
  {
    vkrt_scene_desc d{};
    d.struct_size = sizeof(vkrt_scene_desc);
    d.vertex_count = vertexCount();
    d.positions = m_positions.data();
    d.normals = m_normals.data();
    d.tangents = m_tangents.data();
    d.texcoords0 = m_texcoords0.data();
    d.indices = m_indices.data();
    d.index_count = (uint32_t)m_indices.size();
    d.prim_mesh_count = (uint32_t)m_primMeshes.size();
    d.prim_meshes = m_primMeshes.data();
    d.materials = m_materials.data();
    d.material_count = (uint32_t)m_materials.size();
    d.light_count = (uint32_t)m_lights.size();
    d.lights = m_lights.data();
    d.nodes = m_nodes.data();
    d.node_count = (uint32_t)m_nodes.size();
    tex.clear();
    for(const auto& t : m_textures)
      tex.push_back(vkrt_texture{t.width, t.height, t.rgba.data(), t.srgb ? 1 : 0});
    d.texture_count = (uint32_t)tex.size();
    d.textures = tex.empty() ? nullptr : tex.data();
    return d;
  }
};

// Loads .gltf (JSON + external/embedded buffers) or .glb.  Throws std::runtime_error on malformed
// input (the reference asserts, hello_vulkan.cpp:336,341).
GltfScene loadGltf(const std::string& filename);

// Image decode used by the loader: PNG (8-bit, non-interlaced) natively; any other format through a
// raw sidecar "<image file>.rgba8" (u32 width, u32 height, then RGBA8 rows) written by
// tools/decode_textures.py.  Returns false when the image cannot be decoded.
bool decodeImageFile(const std::string& path, TextureImage& out, std::string& why);
bool decodePngMemory(const uint8_t* data, size_t size, TextureImage& out, std::string& why);
// baseline / extended-sequential / progressive Huffman JPEG, 8-bit, grey or YCbCr (jpeg_decode.cpp)
bool decodeJpegMemory(const uint8_t* data, size_t size, TextureImage& out, std::string& why);
// by signature: PNG or JPEG
bool decodeImageMemory(const uint8_t* data, size_t size, TextureImage& out, std::string& why);

}  // namespace vkrt_host
