"""Deterministic synthetic inputs for the hot path, shared by tests and bench.py.

* conventions: D3D/Vulkan (NDC z in [0,1], UV origin top-left), left-handed view space (+Z forward), row-major
  matrices with row-vector multiplication, non-reversed depth (background = 1.0), perspective FOV 60 deg, near 0.1, far 100;
* geometry-consistent G-buffer: analytic ground plane y=0 + 64 spheres, ray-cast per pixel -> depth, world normal,
  base colour, (roughness, metallic), NDC motion vectors w.r.t. the previous camera (incl. the Halton TAA jitter of
  TemporalAntiAliasing::GetJitterOffset), previous-frame depth;
* camera orbits the scene centre by 0.5 deg per frame.

Everything is torch (runs on CPU for tests and on the GPU for the bench); nothing here is part of the product path."""
import math

import numpy as np
import torch

from . import binding as B

FOV_Y = math.radians(60.0)
NEAR_Z, FAR_Z = 0.1, 100.0
SCENE_SEED = 0x5EED0001
NUM_SPHERES = 64


# ------------------------------------------------------------------------------------------------ camera
def halton(base, index):
    """TemporalAntiAliasing.cpp:43-56 (float arithmetic as in the reference)."""
    result, f = np.float32(0.0), np.float32(1.0)
    while index > 0:
        f = np.float32(f / np.float32(base))
        result = np.float32(result + f * np.float32(index % base))
        index = int(math.floor(np.float32(index) / np.float32(base)))
    return float(result)


def taa_jitter(frame_index, width, height):
    """TemporalAntiAliasing::GetJitterOffset (TemporalAntiAliasing.cpp:63-78): Halton(2,3), 16-sample cycle, NDC units."""
    i = (frame_index % 16) + 1
    jx = (np.float32(halton(2, i)) - np.float32(0.5)) / (np.float32(0.5) * np.float32(width))
    jy = (np.float32(halton(3, i)) - np.float32(0.5)) / (np.float32(0.5) * np.float32(height))
    return float(jx), float(jy)


def _look_at_lh(eye, target, up=(0.0, 1.0, 0.0)):
    eye, target, up = (np.asarray(v, np.float64) for v in (eye, target, up))
    z = target - eye
    z /= np.linalg.norm(z)
    x = np.cross(up, z)
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    m = np.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2] = x, y, z  # row-vector convention: columns are the camera axes
    m[3, :3] = [-np.dot(x, eye), -np.dot(y, eye), -np.dot(z, eye)]
    return m


def _perspective_lh(fov_y, aspect, zn, zf, jitter=(0.0, 0.0)):
    ys = 1.0 / math.tan(0.5 * fov_y)
    xs = ys / aspect
    m = np.zeros((4, 4))
    m[0, 0], m[1, 1] = xs, ys
    m[2, 2], m[2, 3] = zf / (zf - zn), 1.0
    m[3, 2] = -zn * zf / (zf - zn)
    # TemporalAntiAliasing::GetJitteredProjMatrix (TemporalAntiAliasing.hpp:138-155): perspective => m20/m21
    m[2, 0] += jitter[0]
    m[2, 1] += jitter[1]
    return m


def make_camera(frame_index, width, height, jittered=True, reversed_depth=False) -> B.CameraAttribs:
    """CameraAttribs (BasicStructures.fxh:84-149) of frame `frame_index`: orbit of 0.5 deg/frame around the scene centre.
    reversed_depth: the near plane maps to depth 1 and the far plane to 0 (HnCamera::GetProjectionMatrix(UseReverseDepth), HnCamera.cpp:124-140)."""
    ang = math.radians(0.5 * frame_index)
    centre = np.array([0.0, 1.0, 0.0])
    radius, height_y = 14.0, 4.5
    eye = centre + np.array([radius * math.sin(ang), height_y - centre[1], -radius * math.cos(ang)])
    jitter = taa_jitter(frame_index, width, height) if jittered else (0.0, 0.0)
    view = _look_at_lh(eye, centre)
    proj = _perspective_lh(FOV_Y, width / height, FAR_Z, NEAR_Z, jitter) if reversed_depth else _perspective_lh(FOV_Y, width / height, NEAR_Z, FAR_Z, jitter)
    vp = view @ proj
    cam = B.CameraAttribs()
    cam.f4Position[:] = [*eye.astype(np.float32), 1.0]
    cam.f4ViewportSize[:] = [float(width), float(height), float(np.float32(1.0) / np.float32(width)), float(np.float32(1.0) / np.float32(height))]
    dn, df = (1.0, 0.0) if reversed_depth else (0.0, 1.0)
    cam.fNearPlaneZ, cam.fFarPlaneZ, cam.fNearPlaneDepth, cam.fFarPlaneDepth = NEAR_Z, FAR_Z, dn, df
    cam.fSceneNearZ, cam.fSceneFarZ, cam.fSceneNearDepth, cam.fSceneFarDepth = NEAR_Z, FAR_Z, dn, df
    cam.fHandness = -1.0
    cam.uiFrameIndex = frame_index
    cam.fFocusDistance, cam.fFStop, cam.fFocalLength, cam.fSensorWidth, cam.fSensorHeight, cam.fExposure = 10.0, 5.6, 50.0, 36.0, 24.0, 0.0
    cam.f2Jitter[:] = list(jitter)
    for name, m in (("mView", view), ("mProj", proj), ("mViewProj", vp), ("mViewInv", np.linalg.inv(view)), ("mProjInv", np.linalg.inv(proj)),
                    ("mViewProjInv", np.linalg.inv(vp))):
        getattr(cam, name)[:] = list(m.astype(np.float32).reshape(-1))
    return cam


def cam_mat(cam: B.CameraAttribs, name, device, dtype=torch.float32):
    return torch.tensor(list(getattr(cam, name)), dtype=dtype, device=device).view(4, 4)


# ------------------------------------------------------------------------------------------------ scene
class Scene:
    """Ground plane + 64 spheres with per-object materials."""

    def __init__(self, seed=SCENE_SEED):
        rng = np.random.Generator(np.random.PCG64(seed))
        n = NUM_SPHERES
        self.centres = np.stack([rng.uniform(-10, 10, n), np.zeros(n), rng.uniform(-10, 10, n)], 1)
        self.radii = rng.uniform(0.3, 1.2, n)
        self.centres[:, 1] = self.radii + rng.uniform(0.0, 2.8, n)  # resting on / floating above the plane
        base = rng.uniform(0.04, 0.9, (n + 1, 3))
        metallic = (rng.uniform(0, 1, n + 1) < 0.5).astype(np.float64)
        rough = rng.uniform(0.05, 1.0, n + 1)
        low = rng.uniform(0, 1, n + 1) < 0.3
        rough[low] = rng.uniform(0.02, 0.19, int(low.sum()))  # 30 % below the SSR roughness threshold 0.2
        # object 0 = ground plane: mirror-ish dielectric
        base[0], metallic[0], rough[0] = [0.35, 0.35, 0.38], 0.0, 0.1
        self.base, self.metallic, self.rough = base, metallic, rough


def render_gbuffer(scene: Scene, cam: B.CameraAttribs, prev_cam: B.CameraAttribs, width, height, device, rows=None, also_relative_to=None):
    """Ray-casts the scene for camera `cam` (rows = (y0, y1) restricts to a row band of the full frame).
    Returns a dict of float32 tensors: depth (H,W), normal/base_color/material (H,W,4), motion (H,W,2);
    also_relative_to: a second "previous" camera -> "motion_alt" (the motion vectors of the same pixels had that camera come before)."""
    y0, y1 = rows if rows is not None else (0, height)
    dt = torch.float32
    xs = (torch.arange(width, device=device, dtype=dt) + 0.5) / width
    ys = (torch.arange(y0, y1, device=device, dtype=dt) + 0.5) / height
    ndc_x = (2.0 * xs - 1.0)[None, :].expand(y1 - y0, width)
    ndc_y = (1.0 - 2.0 * ys)[:, None].expand(y1 - y0, width)
    proj = cam_mat(cam, "mProj", device)
    view_inv = cam_mat(cam, "mViewInv", device)
    jx, jy = cam.f2Jitter[0], cam.f2Jitter[1]
    # view-space ray through the (jittered) pixel centre: clip.xy = (x*P00 + z*jx, y*P11 + z*jy), w = z
    dvx = (ndc_x - jx) / proj[0, 0]
    dvy = (ndc_y - jy) / proj[1, 1]
    d_view = torch.stack([dvx, dvy, torch.ones_like(dvx)], -1)  # z = 1 => ray parameter t == view-space z
    d_world = d_view @ view_inv[:3, :3]
    origin = torch.tensor(list(cam.f4Position)[:3], device=device, dtype=dt)

    t_best = torch.full((y1 - y0, width), float("inf"), device=device, dtype=dt)
    obj = torch.zeros((y1 - y0, width), device=device, dtype=torch.long)
    hit = torch.zeros((y1 - y0, width), device=device, dtype=torch.bool)
    # ground plane y = 0
    t_pl = -origin[1] / d_world[..., 1]
    ok = (d_world[..., 1] < 0) & (t_pl > NEAR_Z) & (t_pl < FAR_Z)
    t_best = torch.where(ok, t_pl, t_best)
    hit |= ok
    a = (d_world * d_world).sum(-1)
    for i in range(NUM_SPHERES):
        c = torch.tensor(scene.centres[i], device=device, dtype=dt)
        oc = origin - c
        bq = (d_world * oc).sum(-1)
        cq = (oc * oc).sum() - float(scene.radii[i]) ** 2
        disc = bq * bq - a * cq
        t = (-bq - torch.sqrt(disc.clamp_min(0))) / a
        ok = (disc > 0) & (t > NEAR_Z) & (t < t_best)
        t_best = torch.where(ok, t, t_best)
        obj = torch.where(ok, torch.full_like(obj, i + 1), obj)
        hit |= ok

    t_safe = torch.where(hit, t_best, torch.ones_like(t_best))
    pos = origin + d_world * t_safe[..., None]
    centres = torch.tensor(np.concatenate([[[0, 0, 0]], scene.centres]), device=device, dtype=dt)
    radii = torch.tensor(np.concatenate([[1.0], scene.radii]), device=device, dtype=dt)
    n_sphere = (pos - centres[obj]) / radii[obj][..., None]
    n_plane = torch.tensor([0.0, 1.0, 0.0], device=device, dtype=dt).expand_as(pos)
    normal = torch.where((obj == 0)[..., None], n_plane, n_sphere)
    normal = normal / normal.norm(dim=-1, keepdim=True).clamp_min(1e-8)

    # hardware depth of the hit: (m22*z + m32) / z
    depth = torch.where(hit, (proj[2, 2] * t_safe + proj[3, 2]) / t_safe, torch.full_like(t_safe, float(cam.fFarPlaneDepth)))  # background = far-plane depth
    depth = depth.clamp(0.0, 1.0)

    base = torch.tensor(scene.base, device=device, dtype=dt)[obj]
    rough = torch.tensor(scene.rough, device=device, dtype=dt)[obj]
    metal = torch.tensor(scene.metallic, device=device, dtype=dt)[obj]
    zeros = torch.zeros_like(rough)
    hitf = hit.to(dt)
    base_color = torch.cat([base * hitf[..., None], hitf[..., None]], -1)                     # a = opacity
    normal4 = torch.cat([normal * hitf[..., None], zeros[..., None]], -1)
    material = torch.stack([rough * hitf, metal * hitf, zeros, zeros], -1)                    # (roughness, metallic, 0, 0)

    # motion = (clip - jitter) - (prev_clip - prev_jitter) in NDC (ShaderUtilities.fxh:88-91); static scene, moving camera
    def ndc_unjittered(c):
        vp = cam_mat(c, "mViewProj", device)
        p = torch.cat([pos, torch.ones_like(pos[..., :1])], -1) @ vp
        return p[..., :2] / p[..., 3:4] - torch.tensor([c.f2Jitter[0], c.f2Jitter[1]], device=device, dtype=dt)

    here = ndc_unjittered(cam)
    motion = (here - ndc_unjittered(prev_cam)) * hitf[..., None]
    out = {"depth": depth.contiguous(), "normal": normal4.contiguous(), "base_color": base_color.contiguous(), "material": material.contiguous(),
           "motion": motion.contiguous()}
    if also_relative_to is not None:
        out["motion_alt"] = ((here - ndc_unjittered(also_relative_to)) * hitf[..., None]).contiguous()
    return out


def make_frame(scene, frame_index, width, height, device, rows=None, reversed_depth=False):
    """G-buffer of frame `frame_index` plus the previous frame's depth and both cameras."""
    cam = make_camera(frame_index, width, height, reversed_depth=reversed_depth)
    prev = make_camera(max(frame_index - 1, 0), width, height, reversed_depth=reversed_depth)
    g = render_gbuffer(scene, cam, prev, width, height, device, rows)
    gp = render_gbuffer(scene, prev, prev, width, height, device, rows)
    g["prev_depth"] = gp["depth"]
    g["camera"], g["prev_camera"] = cam, prev
    return g


# ------------------------------------------------------------------------------------------------ lights
def make_lights() -> B.PBRShadeAttribs:
    a = B.PBRShadeAttribs()
    a.IBLScale[:] = [1.0, 1.0, 1.0, 1.0]
    a.OcclusionStrength, a.EmissionScale, a.PrefilteredCubeLastMip = 1.0, 1.0, 8.0
    d = np.array([-0.4, -0.8, -0.45])
    d /= np.linalg.norm(d)
    lights = [B.PBRLightAttribs(1, 0, 0, 0, *d.astype(np.float32), -1, 3.0, 3.0, 3.0, 0.0, 0.0, 0.0, 0.0, 0.0)]
    rng = np.random.Generator(np.random.PCG64(SCENE_SEED + 1))
    for _ in range(3):
        p = [rng.uniform(-8, 8), rng.uniform(2, 5), rng.uniform(-8, 8)]
        col = rng.uniform(5, 25, 3)
        lights.append(B.PBRLightAttribs(2, *map(float, p), 0.0, -1.0, 0.0, -1, *map(float, col), 15.0 ** 4, 0.0, 0.0, 0.0, 0.0))
    a.LightCount = len(lights)
    for i, l in enumerate(lights):
        a.Lights[i] = l
    return a


# ------------------------------------------------------------------------------------------------ config-1 HDR buffer
def make_hdr_buffer(width, height, device, seed=1234):
    """1920x1080-style HDR float4 buffer: luminance 2^U(-8,8), random unit-sum chroma, alpha 1, with exact 0 / 1e-12 / 1e4 outliers."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    lum = torch.exp2(torch.rand(height, width, generator=g) * 16.0 - 8.0)
    chroma = torch.rand(height, width, 3, generator=g) + 1e-3
    chroma = chroma / chroma.sum(-1, keepdim=True)
    w = torch.tensor([0.212671, 0.715160, 0.072169])
    rgb = chroma * (lum / (chroma * w).sum(-1))[..., None]
    img = torch.cat([rgb, torch.ones(height, width, 1)], -1).to(torch.float32)
    img[0, 0, :3] = 0.0
    img[0, 1 % width, :3] = 1e-12
    img[0, 2 % width, :3] = 1e4
    return img.contiguous().to(device)


# ------------------------------------------------------------------------------------------------ procedural HDR sky (cube faces, D3D order)
def cube_dirs(size, device):
    """Unit directions of the texel centres of a size x size cube, shape (6, size, size, 3); face order +X,-X,+Y,-Y,+Z,-Z."""
    t = (torch.arange(size, device=device, dtype=torch.float32) + 0.5) / size * 2.0 - 1.0
    tc, sc = torch.meshgrid(t, t, indexing="ij")  # tc: rows (v), sc: columns (u)
    one = torch.ones_like(sc)
    faces = [
        torch.stack([one, -tc, -sc], -1), torch.stack([-one, -tc, sc], -1),
        torch.stack([sc, one, tc], -1), torch.stack([sc, -one, -tc], -1),
        torch.stack([sc, -tc, one], -1), torch.stack([-sc, -tc, -one], -1),
    ]
    d = torch.stack(faces, 0)
    return d / d.norm(dim=-1, keepdim=True)


def sky_radiance(d):
    """Vertical gradient + Gaussian sun lobe (peak 50 000), ground albedo below the horizon. d: (..., 3) unit vectors."""
    up = d[..., 1]
    horizon = torch.tensor([0.9, 0.95, 1.0], device=d.device)
    zenith = torch.tensor([0.15, 0.35, 0.9], device=d.device)
    ground = torch.tensor([0.12, 0.11, 0.10], device=d.device)
    t = up.clamp(0, 1)[..., None] ** 0.5
    sky = horizon * (1 - t) + zenith * t
    col = torch.where((up >= 0)[..., None], sky, ground.expand_as(sky))
    sun = torch.tensor([0.4, 0.8, 0.45], device=d.device)
    sun = sun / sun.norm()
    cosang = (d * sun).sum(-1).clamp(-1, 1)
    lobe = 50000.0 * torch.exp(-(torch.acos(cosang) / 0.02) ** 2)
    return col + lobe[..., None] * torch.tensor([1.0, 0.95, 0.85], device=d.device)


def make_sky_cube(size, device):
    """(6*size, size, 4) float32: faces stacked vertically, as mifx_cubemap expects."""
    rad = sky_radiance(cube_dirs(size, device))
    img = torch.cat([rad, torch.ones_like(rad[..., :1])], -1)
    return img.reshape(6 * size, size, 4).contiguous()
