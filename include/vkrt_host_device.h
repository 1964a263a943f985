/*
 * vkrt_host_device.h -- host/device layout contract of the ray-tracing path.
 *
 * Byte-for-byte mirror of the POD structs that the reference compiles into both its
 * C++ host and its GLSL shaders (reference: shaders/host_device.h:45-137).  A host
 * application that fills the reference's structs can hand the same memory to this
 * library: field order, sizes and offsets are identical (GLSL "scalar" block layout,
 * which equals plain C packing of 4-byte members).
 *
 * Plain C99/C++; no dependencies.  Every layout is pinned by a static assertion.
 */
#ifndef VKRT_HOST_DEVICE_H
#define VKRT_HOST_DEVICE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
#define VKRT_STATIC_ASSERT(c, m) static_assert(c, m)
#else
#define VKRT_STATIC_ASSERT(c, m) _Static_assert(c, m)
#endif

/* Descriptor binding numbers (reference: host_device.h:45-49 SceneBindings,
 * :51-63 RtxBindings).  Kept only as documentation of which reference resource a
 * C-ABI argument replaces; HIP has no descriptor sets. */
enum vkrt_scene_bindings { VKRT_eGlobals = 0, VKRT_eSceneDesc = 1, VKRT_eTextures = 2 };
enum vkrt_rtx_bindings {
  VKRT_eTlas = 0, VKRT_eOutImage = 1, VKRT_ePrimLookup = 2, VKRT_ePosMap = 3,
  VKRT_eNormMap = 4, VKRT_eAccumMap = 5, VKRT_eRoughMap = 6, VKRT_eInMV = 7,
  VKRT_eInNormRough = 8, VKRT_eInViewZ = 9, VKRT_eInRadHitD = 10
};

/* Column-major 4x4 (GLSL mat4 / nvmath::mat4f): element (row r, col c) = m[c*4 + r]. */
typedef struct vkrt_mat4 { float m[16]; } vkrt_mat4;

/* reference: host_device.h:68-73 (uniform buffer written each frame by
 * HelloVulkan::updateUniformBuffer, hello_vulkan.cpp:61-102). */
typedef struct GlobalUniforms {
  vkrt_mat4 viewProj;
  vkrt_mat4 viewInverse;
  vkrt_mat4 projInverse;
} GlobalUniforms;

/* reference: host_device.h:76-84 (raster path only; carried for completeness). */
typedef struct PushConstantRaster {
  vkrt_mat4 modelMatrix;
  vkrt_mat4 inverseTransposeMatrix;
  vkrt_mat4 viewMatrix;
  uint32_t  objIndex;
  int32_t   materialId;
  int32_t   lightsCount;
} PushConstantRaster;

/* reference: host_device.h:88-98 (pushed by HelloVulkan::pathtrace,
 * hello_vulkan.cpp:1440-1442; read by rgen/rchit/rmiss through raycommon.glsl:6). */
typedef struct PushConstantRay {
  float   clearColor[4];
  int32_t frame;
  int32_t lightsCount;
  int32_t samples;
  int32_t depth;
  int32_t useShadows;
  int32_t useAO;
  int32_t useGI;
} PushConstantRay;

/* reference: host_device.h:100-105 (filled at hello_vulkan.cpp:363-368, indexed by
 * gl_InstanceCustomIndexEXT in raytrace.rchit:34). */
typedef struct PrimMeshInfo {
  uint32_t indexOffset;
  uint32_t vertexOffset;
  int32_t  materialIndex;
} PrimMeshInfo;

/* reference: host_device.h:107-117 (eight 64-bit buffer device addresses,
 * hello_vulkan.cpp:370-379). */
typedef struct SceneDesc {
  uint64_t vertexAddress;
  uint64_t normalAddress;
  uint64_t tangentAddress;
  uint64_t uvAddress;
  uint64_t indexAddress;
  uint64_t materialAddress;
  uint64_t lightAddress;
  uint64_t primInfoAddress;
} SceneDesc;

/* reference: host_device.h:119-129 (filled by loadGltfMaterials, hello_vulkan.cpp:207-224). */
typedef struct GltfPBRMaterial {
  float   pbrBaseColorFactor[4];
  int32_t pbrBaseColorTexture;
  float   metallicFactor;
  float   roughnessFactor;
  int32_t metallicRoughnessTexture;
  int32_t normalTexture;
  float   emissiveFactor[3];
  int32_t emissiveTexture;
} GltfPBRMaterial;

/* reference: host_device.h:131-137 (filled by loadGltfLights, hello_vulkan.cpp:226-325).
 * type: 0 point, 1 directional, 2 spot. */
typedef struct GltfLight {
  float   position[3];
  float   color[3];
  float   intensity;
  int32_t type;
} GltfLight;

/* reference: hello_vulkan.h:170-176 (post pass; display only, carried for the CLI). */
typedef struct PushConstantPost {
  float   aspectRatio;
  int32_t rtMode;
  int32_t viewAccumulated;
  int32_t useGI;
} PushConstantPost;

VKRT_STATIC_ASSERT(sizeof(GlobalUniforms) == 192, "GlobalUniforms");
VKRT_STATIC_ASSERT(offsetof(GlobalUniforms, viewInverse) == 64, "GlobalUniforms.viewInverse");
VKRT_STATIC_ASSERT(offsetof(GlobalUniforms, projInverse) == 128, "GlobalUniforms.projInverse");
VKRT_STATIC_ASSERT(sizeof(PushConstantRaster) == 204, "PushConstantRaster");
VKRT_STATIC_ASSERT(offsetof(PushConstantRaster, objIndex) == 192, "PushConstantRaster.objIndex");
VKRT_STATIC_ASSERT(sizeof(PushConstantRay) == 44, "PushConstantRay");
VKRT_STATIC_ASSERT(offsetof(PushConstantRay, frame) == 16, "PushConstantRay.frame");
VKRT_STATIC_ASSERT(offsetof(PushConstantRay, lightsCount) == 20, "PushConstantRay.lightsCount");
VKRT_STATIC_ASSERT(offsetof(PushConstantRay, samples) == 24, "PushConstantRay.samples");
VKRT_STATIC_ASSERT(offsetof(PushConstantRay, depth) == 28, "PushConstantRay.depth");
VKRT_STATIC_ASSERT(offsetof(PushConstantRay, useGI) == 40, "PushConstantRay.useGI");
VKRT_STATIC_ASSERT(sizeof(PrimMeshInfo) == 12, "PrimMeshInfo");
VKRT_STATIC_ASSERT(sizeof(SceneDesc) == 64, "SceneDesc");
VKRT_STATIC_ASSERT(sizeof(GltfPBRMaterial) == 52, "GltfPBRMaterial");
VKRT_STATIC_ASSERT(offsetof(GltfPBRMaterial, pbrBaseColorTexture) == 16, "GltfPBRMaterial.bct");
VKRT_STATIC_ASSERT(offsetof(GltfPBRMaterial, metallicFactor) == 20, "GltfPBRMaterial.metallic");
VKRT_STATIC_ASSERT(offsetof(GltfPBRMaterial, roughnessFactor) == 24, "GltfPBRMaterial.roughness");
VKRT_STATIC_ASSERT(offsetof(GltfPBRMaterial, metallicRoughnessTexture) == 28, "GltfPBRMaterial.mrt");
VKRT_STATIC_ASSERT(offsetof(GltfPBRMaterial, normalTexture) == 32, "GltfPBRMaterial.nt");
VKRT_STATIC_ASSERT(offsetof(GltfPBRMaterial, emissiveFactor) == 36, "GltfPBRMaterial.emissive");
VKRT_STATIC_ASSERT(offsetof(GltfPBRMaterial, emissiveTexture) == 48, "GltfPBRMaterial.et");
VKRT_STATIC_ASSERT(sizeof(GltfLight) == 32, "GltfLight");
VKRT_STATIC_ASSERT(offsetof(GltfLight, color) == 12, "GltfLight.color");
VKRT_STATIC_ASSERT(offsetof(GltfLight, intensity) == 24, "GltfLight.intensity");
VKRT_STATIC_ASSERT(offsetof(GltfLight, type) == 28, "GltfLight.type");
VKRT_STATIC_ASSERT(sizeof(PushConstantPost) == 16, "PushConstantPost");

#endif /* VKRT_HOST_DEVICE_H */
