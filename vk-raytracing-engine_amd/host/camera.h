// camera.h -- camera matrices feeding raytrace.rgen through GlobalUniforms.
//
// Host-side replacement for what HelloVulkan::updateUniformBuffer (reference
// hello_vulkan.cpp:61-102) takes from nvpro_core: CameraManip.getMatrix()/getFov(),
// nvmath::perspectiveVK(fov, aspect, 0.1, 1000) and nvmath::invert.  nvpro_core is not in the
// reference tree; behaviour restated per SURVEY.md Appendix D (right-handed look-at, Vulkan clip
// space with depth 0..1 and Y flipped, default vertical fov 60 degrees).  Column-major storage as
// GLSL mat4 / nvmath::mat4f: element (row r, col c) = m[c*4 + r].
#pragma once
#include <cmath>
#include <cstring>
#include "../../include/vkrt_host_device.h"

namespace vkrt_host {

struct Vec3
{
  float x = 0, y = 0, z = 0;
};
inline Vec3 sub(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline float dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Vec3 cross(Vec3 a, Vec3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline Vec3 normalize(Vec3 a)
{
  const float l = std::sqrt(dot(a, a));
  return {a.x / l, a.y / l, a.z / l};
}

inline vkrt_mat4 identity()
{
  vkrt_mat4 r;
  std::memset(&r, 0, sizeof r);
  r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0f;
  return r;
}
inline float& at(vkrt_mat4& M, int row, int col) { return M.m[col * 4 + row]; }
inline float at(const vkrt_mat4& M, int row, int col) { return M.m[col * 4 + row]; }

inline vkrt_mat4 mul(const vkrt_mat4& A, const vkrt_mat4& B)
{
  vkrt_mat4 R;
  for(int c = 0; c < 4; c++)
    for(int r = 0; r < 4; r++)
    {
      float s = 0.f;
      for(int k = 0; k < 4; k++)
        s += at(A, r, k) * at(B, k, c);
      at(R, r, c) = s;
    }
  return R;
}

// Right-handed look-at view matrix (CameraManip.getMatrix()).
inline vkrt_mat4 lookAt(Vec3 eye, Vec3 center, Vec3 up)
{
  const Vec3 f = normalize(sub(center, eye));
  const Vec3 s = normalize(cross(f, up));
  const Vec3 u = cross(s, f);
  vkrt_mat4 M = identity();
  at(M, 0, 0) = s.x; at(M, 0, 1) = s.y; at(M, 0, 2) = s.z; at(M, 0, 3) = -dot(s, eye);
  at(M, 1, 0) = u.x; at(M, 1, 1) = u.y; at(M, 1, 2) = u.z; at(M, 1, 3) = -dot(u, eye);
  at(M, 2, 0) = -f.x; at(M, 2, 1) = -f.y; at(M, 2, 2) = -f.z; at(M, 2, 3) = dot(f, eye);
  return M;
}

// nvmath::perspectiveVK(fovy_deg, aspect, n, f): RH, depth 0..1, Y flipped (launch row 0 = image top).
inline vkrt_mat4 perspectiveVK(float fovyDeg, float aspect, float n, float f)
{
  const float t = n * std::tan(fovyDeg * 0.017453292519943295f * 0.5f);
  const float b = -t, l = b * aspect, r = t * aspect;
  vkrt_mat4 M;
  std::memset(&M, 0, sizeof M);
  at(M, 0, 0) = (2.f * n) / (r - l);
  at(M, 1, 1) = -(2.f * n) / (t - b);
  at(M, 0, 2) = (r + l) / (r - l);
  at(M, 1, 2) = (t + b) / (t - b);
  at(M, 2, 2) = f / (n - f);
  at(M, 3, 2) = -1.f;
  at(M, 2, 3) = (f * n) / (n - f);
  return M;
}

// General 4x4 inverse by cofactors (nvmath::invert).
inline vkrt_mat4 invert(const vkrt_mat4& A)
{
  const float* a = A.m;
  float inv[16];
  inv[0] = a[5] * a[10] * a[15] - a[5] * a[11] * a[14] - a[9] * a[6] * a[15] + a[9] * a[7] * a[14] + a[13] * a[6] * a[11] - a[13] * a[7] * a[10];
  inv[4] = -a[4] * a[10] * a[15] + a[4] * a[11] * a[14] + a[8] * a[6] * a[15] - a[8] * a[7] * a[14] - a[12] * a[6] * a[11] + a[12] * a[7] * a[10];
  inv[8] = a[4] * a[9] * a[15] - a[4] * a[11] * a[13] - a[8] * a[5] * a[15] + a[8] * a[7] * a[13] + a[12] * a[5] * a[11] - a[12] * a[7] * a[9];
  inv[12] = -a[4] * a[9] * a[14] + a[4] * a[10] * a[13] + a[8] * a[5] * a[14] - a[8] * a[6] * a[13] - a[12] * a[5] * a[10] + a[12] * a[6] * a[9];
  inv[1] = -a[1] * a[10] * a[15] + a[1] * a[11] * a[14] + a[9] * a[2] * a[15] - a[9] * a[3] * a[14] - a[13] * a[2] * a[11] + a[13] * a[3] * a[10];
  inv[5] = a[0] * a[10] * a[15] - a[0] * a[11] * a[14] - a[8] * a[2] * a[15] + a[8] * a[3] * a[14] + a[12] * a[2] * a[11] - a[12] * a[3] * a[10];
  inv[9] = -a[0] * a[9] * a[15] + a[0] * a[11] * a[13] + a[8] * a[1] * a[15] - a[8] * a[3] * a[13] - a[12] * a[1] * a[11] + a[12] * a[3] * a[9];
  inv[13] = a[0] * a[9] * a[14] - a[0] * a[10] * a[13] - a[8] * a[1] * a[14] + a[8] * a[2] * a[13] + a[12] * a[1] * a[10] - a[12] * a[2] * a[9];
  inv[2] = a[1] * a[6] * a[15] - a[1] * a[7] * a[14] - a[5] * a[2] * a[15] + a[5] * a[3] * a[14] + a[13] * a[2] * a[7] - a[13] * a[3] * a[6];
  inv[6] = -a[0] * a[6] * a[15] + a[0] * a[7] * a[14] + a[4] * a[2] * a[15] - a[4] * a[3] * a[14] - a[12] * a[2] * a[7] + a[12] * a[3] * a[6];
  inv[10] = a[0] * a[5] * a[15] - a[0] * a[7] * a[13] - a[4] * a[1] * a[15] + a[4] * a[3] * a[13] + a[12] * a[1] * a[7] - a[12] * a[3] * a[5];
  inv[14] = -a[0] * a[5] * a[14] + a[0] * a[6] * a[13] + a[4] * a[1] * a[14] - a[4] * a[2] * a[13] - a[12] * a[1] * a[6] + a[12] * a[2] * a[5];
  inv[3] = -a[1] * a[6] * a[11] + a[1] * a[7] * a[10] + a[5] * a[2] * a[11] - a[5] * a[3] * a[10] - a[9] * a[2] * a[7] + a[9] * a[3] * a[6];
  inv[7] = a[0] * a[6] * a[11] - a[0] * a[7] * a[10] - a[4] * a[2] * a[11] + a[4] * a[3] * a[10] + a[8] * a[2] * a[7] - a[8] * a[3] * a[6];
  inv[11] = -a[0] * a[5] * a[11] + a[0] * a[7] * a[9] + a[4] * a[1] * a[11] - a[4] * a[3] * a[9] - a[8] * a[1] * a[7] + a[8] * a[3] * a[5];
  inv[15] = a[0] * a[5] * a[10] - a[0] * a[6] * a[9] - a[4] * a[1] * a[10] + a[4] * a[2] * a[9] + a[8] * a[1] * a[6] - a[8] * a[2] * a[5];
  const float det = a[0] * inv[0] + a[1] * inv[4] + a[2] * inv[8] + a[3] * inv[12];
  const float id = 1.0f / det;
  vkrt_mat4 R;
  for(int i = 0; i < 16; i++)
    R.m[i] = inv[i] * id;
  return R;
}

// The slice of nvh::CameraManipulator the ray-tracing path uses (main.cpp:158-160,
// hello_vulkan.cpp:66-67, :1506-1520): look-at state + vertical fov, default 60 degrees.
struct CameraManipulator
{
  Vec3 eye{0, 0, 15}, center{0, 0, 0}, up{0, 1, 0};  // main.cpp:160
  float fov = 60.0f;
  int width = 1280, height = 720;
  void setWindowSize(int w, int h) { width = w; height = h; }
  void setLookat(Vec3 e, Vec3 c, Vec3 u) { eye = e; center = c; up = u; }
  void setFov(float f) { fov = f; }
  float getFov() const { return fov; }
  vkrt_mat4 getMatrix() const { return lookAt(eye, center, up); }
};

// HelloVulkan::updateUniformBuffer (hello_vulkan.cpp:61-72): the host UBO contents.
inline GlobalUniforms makeGlobalUniforms(const CameraManipulator& cam, int width, int height)
{
  const float aspect = width / static_cast<float>(height);
  const vkrt_mat4 view = cam.getMatrix();
  const vkrt_mat4 proj = perspectiveVK(cam.getFov(), aspect, 0.1f, 1000.0f);
  GlobalUniforms u;
  u.viewProj = mul(proj, view);
  u.viewInverse = invert(view);
  u.projInverse = invert(proj);
  return u;
}

}  // namespace vkrt_host
