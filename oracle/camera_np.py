"""numpy restatement of the camera matrices the reference feeds raytrace.rgen (TEST
INFRASTRUCTURE).  Reference: HelloVulkan::updateUniformBuffer hello_vulkan.cpp:61-102 using
nvmath::perspectiveVK / nvmath::invert / CameraManip (nvpro_core, not in the reference tree;
behaviour per SURVEY.md Appendix D: right-handed, depth 0..1, Y flipped, default fov 60 deg,
look-at view matrix).  PARITY UNPINNED (third-party math, no reference test)."""
import numpy as np


def look_at(eye, center, up):
    eye, center, up = (np.asarray(v, np.float64) for v in (eye, center, up))
    f = center - eye
    f /= np.linalg.norm(f)
    s = np.cross(f, up)
    s /= np.linalg.norm(s)
    u = np.cross(s, f)
    M = np.eye(4)
    M[0, :3], M[1, :3], M[2, :3] = s, u, -f
    M[0, 3], M[1, 3], M[2, 3] = -s @ eye, -u @ eye, f @ eye
    return M


def perspective_vk(fovy_deg, aspect, near, far):
    t = np.tan(np.radians(fovy_deg) * 0.5)
    M = np.zeros((4, 4))
    M[0, 0] = 1.0 / (aspect * t)
    M[1, 1] = -1.0 / t
    M[2, 2] = far / (near - far)
    M[2, 3] = (far * near) / (near - far)
    M[3, 2] = -1.0
    return M


def global_uniforms(eye=(0, 0, 15), center=(0, 0, 0), up=(0, 1, 0), fov=60.0, width=1280, height=720, near=0.1, far=1000.0):
    """Defaults: main.cpp:158-160 (lookat), CameraManip fov 60, hello_vulkan.cpp:67 (near/far)."""
    view = look_at(eye, center, up)
    proj = perspective_vk(fov, width / float(height), near, far)
    return (proj @ view).astype(np.float32), np.linalg.inv(view).astype(np.float32), np.linalg.inv(proj).astype(np.float32)
