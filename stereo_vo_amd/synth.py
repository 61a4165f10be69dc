"""Deterministic synthetic stereo sequences: textured planes seen by a rectified pinhole pair.

Three scene types: "planes" (a far wall, the ground and three or four facades: the default, and what every committed number before
round 3 was measured on), "relief" (the same plus a few dozen small billboards at depths of 4 .. 22 m: depth discontinuities
everywhere in the image, so that the scene is not a handful of planes -- a near-degenerate configuration for a fundamental matrix)
and "street" (round 4: the ground, a far wall beyond the end of the trajectory and a facade every 4.5 m on alternating sides all
along it -- a world long enough for hundreds of frames of forward motion, bench.py's non-repeating sequences).

Plays the role of the image source of the reference's demo (demo-stereo-odometry/demo-main.cpp:200-220,
mrpt CCameraSensor) on inputs that can be generated on the GPU box from this repository alone.
Rendering uses torch so that it runs on the CPU here and on cuda:0 in bench.py; every random quantity comes
from a seeded xorshift64* so that a (seed, frame) pair always names the same scene and motion.
"""
import math
import numpy as np
import torch

from .abi import StereoCamera

MASK64 = (1 << 64) - 1


class XorShift64Star:
    def __init__(self, seed):
        self.s = (seed & MASK64) or 0x9E3779B97F4A7C15

    def next(self):
        x = self.s
        x ^= x >> 12
        x ^= (x << 25) & MASK64
        x ^= x >> 27
        self.s = x
        return (x * 0x2545F4914F6CDD1D) & MASK64

    def uniform(self, a=0.0, b=1.0):
        return a + (b - a) * ((self.next() >> 11) / float(1 << 53))

    def randint(self, a, b):  # inclusive
        return a + int(self.next() % (b - a + 1))


def _manhattan_texture(rng, size, n_rect):
    """Axis-aligned rectangles 4..40 texels with intensities U{16..240} over a mid-gray canvas."""
    tex = np.full((size, size), 128, np.uint8)
    for _ in range(n_rect):
        rw, rh = rng.randint(4, 40), rng.randint(4, 40)
        x0, y0 = rng.randint(0, size - rw), rng.randint(0, size - rh)
        tex[y0:y0 + rh, x0:x0 + rw] = rng.randint(16, 240)
    # fine detail so that corner neighbourhoods are distinctive (small rectangles 2..8 texels)
    for _ in range(3 * n_rect):
        rw, rh = rng.randint(2, 8), rng.randint(2, 8)
        x0, y0 = rng.randint(0, size - rw), rng.randint(0, size - rh)
        tex[y0:y0 + rh, x0:x0 + rw] = rng.randint(16, 240)
    return tex


def _rot_zyx(yaw, pitch, roll):
    cy, sy, cp, sp, cr, sr = math.cos(yaw), math.sin(yaw), math.cos(pitch), math.sin(pitch), math.cos(roll), math.sin(roll)
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1.0]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    return Rz @ Ry @ Rx


class SyntheticStereoWorld:
    """A static world of textured planes and a camera trajectory.

    Camera frame: x right, y down, z forward.  World frame = frame of the left camera at t = 0.
    pose[t] is the 4x4 camera-to-world transform of the left camera at frame t; the right camera sits at
    +baseline along the left camera's x axis (S5:185: X2c = X1c - baseline).
    """

    _scene_cache = {}

    def __init__(self, width, height, focal, baseline=0.12, seed=0, n_frames=20, device="cpu",
                 cx=None, cy=None, noise_sigma=2.0, tex_size=2048, scene_seed=None, scene="planes", noise_on_device=False):
        assert scene in ("planes", "relief", "street")
        # noise_on_device: the pixel noise comes from torch's generator of `device` (seeded per eye) instead of the CPU generator --
        # ~6x faster rendering on a GPU; the images then depend on the device's generator, so tests keep the default
        self.noise_on_device = bool(noise_on_device)
        self.scene = scene
        self.w, self.h, self.f, self.B = int(width), int(height), float(focal), float(baseline)
        self.cx = (self.w - 1) / 2.0 if cx is None else float(cx)
        self.cy = (self.h - 1) / 2.0 if cy is None else float(cy)
        self.seed = int(seed)
        self.n_frames = int(n_frames)
        self.device = torch.device(device)
        self.noise_sigma = float(noise_sigma)
        # scene (planes + textures) and trajectory have separate seeds so that many independent streams can share
        # one scene's textures (they are the slow part to mint); by default both follow `seed`.
        self.scene_seed = self.seed if scene_seed is None else int(scene_seed)
        self._build_scene(tex_size)
        self._build_trajectory()
        ys, xs = torch.meshgrid(torch.arange(self.h, device=self.device, dtype=torch.float32),
                                torch.arange(self.w, device=self.device, dtype=torch.float32), indexing="ij")
        self._ray = torch.stack([(xs - self.cx) / self.f, (ys - self.cy) / self.f, torch.ones_like(xs)], dim=-1)  # h,w,3

    def _build_scene(self, tex_size):
        self.length = 0.175 * self.n_frames                 # expected forward travel (0.05 .. 0.3 m per frame)
        key = (self.scene_seed, tex_size, self.scene) + ((int(self.length),) if self.scene == "street" else ())
        if key in SyntheticStereoWorld._scene_cache:
            self.planes, texs = SyntheticStereoWorld._scene_cache[key]
            self.textures = [torch.from_numpy(t).to(self.device).float() for t in texs]
            self.tex_size = tex_size
            return
        rng = XorShift64Star(0x5EED0000 + self.scene_seed)
        # planes: (point, normal, e_u, e_v, (umin, umax, vmin, vmax), metres per texel)
        planes = []
        # far wall
        planes.append((np.array([0, 0, 26.0]), np.array([0, 0, -1.0]), np.array([1.0, 0, 0]), np.array([0, 1.0, 0]), (-60, 60, -40, 40), 0.06))
        # ground
        planes.append((np.array([0, 1.7, 0]), np.array([0, -1.0, 0]), np.array([1.0, 0, 0]), np.array([0, 0, 1.0]), (-40, 40, 0.5, 80), 0.04))
        if self.scene == "street":
            # the same kind of world, long: far wall beyond the end of the trajectory, ground all along, a facade every 4.5 m on
            # alternating sides (texel size grows with the distance it is first seen from; textures are shared, at random offsets)
            L = self.length
            planes[0] = (np.array([0, 0, L + 45.0]), np.array([0, 0, -1.0]), np.array([1.0, 0, 0]), np.array([0, 1.0, 0]), (-120, 120, -60, 60), 0.0022 * (L + 45.0))
            planes[1] = (np.array([0, 1.7, 0]), np.array([0, -1.0, 0]), np.array([1.0, 0, 0]), np.array([0, 0, 1.0]), (-60, 60, 0.5, L + 60.0), 0.04)
            n_base = 6
            k = 0
            z = 6.0
            while z < L + 32.0:
                side = 1.0 if k % 2 == 0 else -1.0
                half = rng.uniform(1.2, 3.0)
                xc = side * (half + rng.uniform(0.6, 4.0))         # clear of the camera's path
                tilt = rng.uniform(-0.35, 0.35)
                n = np.array([math.sin(tilt), 0, -math.cos(tilt)])
                eu = np.array([math.cos(tilt), 0, math.sin(tilt)])
                ev = np.array([0, 1.0, 0])
                mpt = max(0.012, 0.0011 * z)
                if k < n_base - 2:
                    planes.append((np.array([xc, 0, z]), n, eu, ev, (-half, half, -3.5, 1.7), mpt))
                else:
                    planes.append((np.array([xc, 0, z]), n, eu, ev, (-half, half, -3.5, 1.7), mpt, 2 + rng.randint(0, n_base - 3), (rng.uniform(0, tex_size), rng.uniform(0, tex_size))))
                z += rng.uniform(3.5, 5.5)
                k += 1
            n_tex = n_base
        else:
            # a few nearer facades with small random tilt
            n_fac = 3 + rng.randint(0, 1)
            for k in range(n_fac):
                z = rng.uniform(7.0, 16.0)
                xc = rng.uniform(-7.0, 7.0)
                half = rng.uniform(1.2, 3.0)
                tilt = rng.uniform(-0.35, 0.35)
                n = np.array([math.sin(tilt), 0, -math.cos(tilt)])
                eu = np.array([math.cos(tilt), 0, math.sin(tilt)])
                ev = np.array([0, 1.0, 0])
                planes.append((np.array([xc, 0, z]), n, eu, ev, (-half, half, -3.5, 1.7), 0.012 + 0.002 * k))
            n_tex = len(planes)
        if self.scene == "relief":
            # billboards: small, nearly fronto-parallel textured rectangles scattered through the viewing volume (the camera
            # advances ~1 m over a sequence).  They share the base planes' textures at random offsets (entries 6, 7 of the tuple)
            # with a texel size that grows with depth, so that every one shows corners at the scale the detector looks at.
            half_fov = 0.5 * self.w / self.f
            for k in range(28):
                z = rng.uniform(4.0, 22.0)
                xc = rng.uniform(-0.9 * half_fov * z, 0.9 * half_fov * z)
                yc = rng.uniform(-0.35 * z, 1.0)
                hw, hh = rng.uniform(0.25, 0.9) * (0.5 + z / 12.0), rng.uniform(0.25, 0.9) * (0.5 + z / 12.0)
                tilt = rng.uniform(-0.25, 0.25)
                n = np.array([math.sin(tilt), 0, -math.cos(tilt)])
                eu = np.array([math.cos(tilt), 0, math.sin(tilt)])
                ev = np.array([0, 1.0, 0])
                planes.append((np.array([xc, yc, z]), n, eu, ev, (-hw, hw, -hh, hh), 0.004 + 0.0009 * z,
                               rng.randint(0, n_tex - 1), (rng.uniform(0, tex_size), rng.uniform(0, tex_size))))
        self.planes = planes
        texs = [_manhattan_texture(rng, tex_size, 5000) for _ in range(n_tex)]
        SyntheticStereoWorld._scene_cache[key] = (planes, texs)
        self.textures = [torch.from_numpy(t).to(self.device).float() for t in texs]
        self.tex_size = tex_size

    def _build_trajectory(self):
        rng = XorShift64Star(0x7AA70000 + self.seed)
        self.poses = [np.eye(4)]
        self.deltas = [np.eye(4)]
        for t in range(1, self.n_frames):
            fwd = rng.uniform(0.05, 0.3)
            yaw_deg = rng.uniform(-0.5, 0.5)                    # rotation about the camera's y (vertical) axis
            if self.scene == "street":
                # Keep to the middle of the street (round 6).  The unbiased +-0.5 deg heading walk of rounds 1-5 drifted the camera 1.4 .. 3 m
                # sideways within ~220 frames, where the facades stand 0.6 .. 4 m off the centre line: 8 - 10 of bench.py's 192 streams walked
                # INTO a facade (tools/stream_loss.py: keypoints 1850 -> 1199 -> 892 -> 70 -> 0 over the last four frames as the facade's
                # texture, seen from under a metre, outgrows the detector's scale) and stayed invalid for the rest of the leg.  A driver steers:
                # a correction towards heading 0 and lateral offset 0, at most half a degree per frame on top of the random part.
                P = self.poses[-1]
                heading = math.degrees(math.atan2(P[0, 2], P[2, 2]))
                yaw_deg += max(-0.5, min(0.5, -(0.12 * heading + 0.35 * P[0, 3])))
            yaw_cam = math.radians(yaw_deg)
            pitch_cam = math.radians(rng.uniform(-0.1, 0.1))
            roll_cam = math.radians(rng.uniform(-0.1, 0.1))
            # axes in camera convention: "yaw" about y-down, "pitch" about x, "roll" about z
            cyw, syw = math.cos(yaw_cam), math.sin(yaw_cam)
            Ry = np.array([[cyw, 0, syw], [0, 1, 0], [-syw, 0, cyw]])
            cpt, spt = math.cos(pitch_cam), math.sin(pitch_cam)
            Rx = np.array([[1, 0, 0], [0, cpt, -spt], [0, spt, cpt]])
            crl, srl = math.cos(roll_cam), math.sin(roll_cam)
            Rz = np.array([[crl, -srl, 0], [srl, crl, 0], [0, 0, 1]])
            D = np.eye(4)
            D[:3, :3] = Ry @ Rx @ Rz
            D[:3, 3] = [rng.uniform(-0.01, 0.01), rng.uniform(-0.005, 0.005), fwd]
            self.deltas.append(D)
            self.poses.append(self.poses[-1] @ D)

    def camera(self) -> StereoCamera:
        return StereoCamera.simple(self.f, self.cx, self.cy, self.B, self.w, self.h)

    def gt_delta(self, t):
        """4x4 pose of frame t with respect to frame t-1 (what result.outPose estimates, S5:715-718)."""
        return self.deltas[t]

    def _render_eye(self, T, gen_seed):
        dev = self.device
        if not hasattr(self, "_plane_dev"):
            self._plane_dev = {}
        R = torch.tensor(T[:3, :3], dtype=torch.float32, device=dev)
        o = torch.tensor(T[:3, 3], dtype=torch.float32, device=dev)
        d = self._ray @ R.T                                        # h,w,3 world directions
        best_s = torch.full((self.h, self.w), float("inf"), device=dev)
        img = torch.full((self.h, self.w), 90.0, device=dev)
        cam_fwd = T[:3, 2]
        Rn, on = T[:3, :3], T[:3, 3]
        for pi, pl in enumerate(self.planes):
            P, n, eu, ev, ext, mpt = pl[:6]
            if self.scene == "street" and pi >= 2 and float(np.dot(np.asarray(P) - T[:3, 3], cam_fwd)) < -4.0:
                continue                                           # a facade the camera has passed: behind it, cannot be seen
            # On a GPU a plane is rendered inside the bounding box of its projected rectangle only (same arithmetic per pixel;
            # the pixels outside cannot pass the extent test): a facade covers a small part of the image and the renderer is
            # bound by the per-plane passes over it.  The CPU path (tests, golden vectors) keeps whole-image passes.
            ys0, ys1, xs0, xs1 = 0, self.h, 0, self.w
            if getattr(self, "_bbox", dev.type != "cpu"):
                cor = np.array([np.asarray(P) + a_ * np.asarray(eu) + b_ * np.asarray(ev) for a_ in (ext[0], ext[1]) for b_ in (ext[2], ext[3])])
                cc = (cor - on) @ Rn                               # camera coordinates of the four corners
                if cc[:, 2].min() > 0.2:
                    u = self.f * cc[:, 0] / cc[:, 2] + self.cx; v = self.f * cc[:, 1] / cc[:, 2] + self.cy
                    xs0, xs1 = max(0, int(math.floor(u.min())) - 2), min(self.w, int(math.ceil(u.max())) + 3)
                    ys0, ys1 = max(0, int(math.floor(v.min())) - 2), min(self.h, int(math.ceil(v.max())) + 3)
                    if xs0 >= xs1 or ys0 >= ys1:
                        continue                                   # wholly outside the image
            tex = self.textures[pl[6] if len(pl) > 6 else pi]
            tu0, tv0 = pl[7] if len(pl) > 7 else (0.0, 0.0)
            # the planes' vectors live on the device once (four blocking host-to-device copies per plane and eye otherwise)
            if pi not in self._plane_dev:
                self._plane_dev[pi] = tuple(torch.tensor(v, dtype=torch.float32, device=dev) for v in (P, n, eu, ev))
            Pt, nt, eut, evt = self._plane_dev[pi]
            ds = d[ys0:ys1, xs0:xs1]
            bs, im = best_s[ys0:ys1, xs0:xs1], img[ys0:ys1, xs0:xs1]
            denom = ds @ nt
            s = torch.dot(nt, Pt - o) / denom
            X = o + s.unsqueeze(-1) * ds - Pt
            a, b = X @ eut, X @ evt
            ok = (s > 0.05) & (s < bs) & (a >= ext[0]) & (a <= ext[1]) & (b >= ext[2]) & (b <= ext[3]) & torch.isfinite(s)
            tu = a / mpt + (self.tex_size / 2.0 + tu0)
            tv = b / mpt + (self.tex_size / 2.0 + tv0)
            # wrap the texture so that large planes stay textured everywhere
            tu = torch.remainder(tu, self.tex_size - 1.0)
            tv = torch.remainder(tv, self.tex_size - 1.0)
            x0 = tu.floor().clamp(0, self.tex_size - 2).long()
            y0 = tv.floor().clamp(0, self.tex_size - 2).long()
            fx, fy = tu - x0, tv - y0
            v = (tex[y0, x0] * (1 - fx) * (1 - fy) + tex[y0, x0 + 1] * fx * (1 - fy)
                 + tex[y0 + 1, x0] * (1 - fx) * fy + tex[y0 + 1, x0 + 1] * fx * fy)
            if (ys1 - ys0, xs1 - xs0) == (self.h, self.w):
                img = torch.where(ok, v, img)
                best_s = torch.where(ok, s, best_s)
            else:
                im.copy_(torch.where(ok, v, im))
                bs.copy_(torch.where(ok, s, bs))
        if self.noise_sigma > 0 and self.noise_on_device and dev.type != "cpu":
            g = torch.Generator(device=dev)
            g.manual_seed(gen_seed)
            img = img + torch.randn((self.h, self.w), generator=g, dtype=torch.float32, device=dev) * self.noise_sigma
        elif self.noise_sigma > 0:
            g = torch.Generator(device="cpu")
            g.manual_seed(gen_seed)
            noise = torch.randn((self.h, self.w), generator=g, dtype=torch.float32) * self.noise_sigma
            img = img + noise.to(dev)
        return img.round().clamp(0, 255).to(torch.uint8)

    def render(self, t):
        """(left, right) uint8 [h, w] tensors on self.device for frame t."""
        T = self.poses[t]
        Tr = T.copy()
        Tr[:3, 3] = T[:3, 3] + T[:3, :3] @ np.array([self.B, 0, 0])
        base = (self.seed * 100003 + t) * 2
        return self._render_eye(T, base + 1), self._render_eye(Tr, base + 2)


def pose6_to_matrix(p):
    """x y z yaw pitch roll (R = Rz(yaw) Ry(pitch) Rx(roll)) -> 4x4."""
    M = np.eye(4)
    M[:3, :3] = _rot_zyx(p[3], p[4], p[5])
    M[:3, 3] = p[:3]
    return M


def pose_error(A, B):
    """(rotation angle [rad], translation norm) between two 4x4 transforms."""
    dR = A[:3, :3].T @ B[:3, :3]
    c = max(-1.0, min(1.0, (np.trace(dR) - 1.0) / 2.0))
    return math.acos(c), float(np.linalg.norm(A[:3, 3] - B[:3, 3]))
