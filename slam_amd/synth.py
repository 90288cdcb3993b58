"""Seeded synthetic inputs for the tracking hot path (numpy only; no dependency on the oracle).

Frames are cut from an unbounded procedural canvas, so frame t of a sequence is frame 0 translated
by an integer (u, v) and the true position of every template is known.
"""
import numpy as np

SEED_BASE = 0x51A40000


def hash32(x, y, seed):
    """Counter-based 32-bit hash of integer coordinates (vectorised, wraps mod 2^32)."""
    x = np.asarray(x).astype(np.uint64)
    y = np.asarray(y).astype(np.uint64)
    h = (x * np.uint64(0x9E3779B1) + y * np.uint64(0x85EBCA77) + np.uint64(seed & 0xFFFFFFFF) * np.uint64(0xC2B2AE3D)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(15)
    h = (h * np.uint64(0x2C1B3C6D)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(12)
    h = (h * np.uint64(0x297A2D39)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(15)
    return h.astype(np.uint32)


def canvas(x0, y0, w, h, seed, cell=16):
    """u8 texture at canvas coordinates [x0, x0+w) x [y0, y0+h): random-intensity blocks of
    `cell` pixels (junctions give FAST corners, no periodicity so correlation peaks are unique),
    finer 5-pixel blocks and per-pixel noise."""
    X = (np.arange(w, dtype=np.int64) + x0 + (1 << 20))[None, :] + np.zeros((h, 1), np.int64)
    Y = (np.arange(h, dtype=np.int64) + y0 + (1 << 20))[:, None] + np.zeros((1, w), np.int64)
    coarse = (hash32(X // cell, Y // cell, seed ^ 0x1234567) % np.uint32(150)).astype(np.int64)
    fine = (hash32(X // 5, Y // 5, seed ^ 0x89ABCDE) % np.uint32(40)).astype(np.int64)
    noise = (hash32(X, Y, seed) % np.uint32(24)).astype(np.int64)
    v = 30 + coarse + fine + noise
    return np.clip(v, 0, 255).astype(np.uint8)


def canvas_smooth(x0, y0, w, h, seed, cell=32):
    """u8 texture at canvas coordinates like `canvas`, but band-limited the way camera images are: value noise (random
    intensities on a `cell`-pixel grid, bilinearly interpolated, integer arithmetic) plus a weaker octave at half the
    cell and a little per-pixel noise.  A pyramid level subsampled by 4 still shows the structure, so a coarse lattice
    search lands next to the truth."""
    X = (np.arange(w, dtype=np.int64) + x0 + (1 << 20))[None, :] + np.zeros((h, 1), np.int64)
    Y = (np.arange(h, dtype=np.int64) + y0 + (1 << 20))[:, None] + np.zeros((1, w), np.int64)

    def octave(c, amp, salt):
        gx, gy, fx, fy = X // c, Y // c, X % c, Y % c
        v = lambda dx, dy: (hash32(gx + dx, gy + dy, seed ^ salt) % np.uint32(amp)).astype(np.int64)
        top = v(0, 0) * (c - fx) + v(1, 0) * fx
        bot = v(0, 1) * (c - fx) + v(1, 1) * fx
        return (top * (c - fy) + bot * fy) // (c * c)

    noise = (hash32(X, Y, seed) % np.uint32(3)).astype(np.int64)
    v = 10 + octave(cell, 200, 0x1234567) + octave(max(cell // 2, 2), 30, 0x89ABCDE) + noise
    return np.clip(v, 0, 255).astype(np.uint8)


def frame(w, h, seed, shift=(0, 0)):
    return canvas(shift[0], shift[1], w, h, seed)


def proptest_image(w, h):
    """The image of the reference's tests/test_matching.cpp:21-23:
    (uchar)((int32)(15485867*x + 15486883*y) % 7907) with C truncating '%'."""
    x = np.arange(w, dtype=np.uint64)[None, :]
    y = np.arange(h, dtype=np.uint64)[:, None]
    v = ((np.uint64(15485867) * x + np.uint64(15486883) * y) & np.uint64(0xFFFFFFFF)).astype(np.int64)
    v = np.where(v >= (1 << 31), v - (1 << 32), v)          # reinterpret as int32
    r = np.fmod(v, 7907)                                     # truncation toward zero
    return (r & 0xFF).astype(np.uint8)


def feature_grid(w, h, n, margin, rng):
    """n feature centres on a jittered grid at least `margin` pixels from the border."""
    cols = int(np.ceil(np.sqrt(n * w / h)))
    rows = int(np.ceil(n / cols))
    xs = np.linspace(margin, w - 1 - margin, cols)
    ys = np.linspace(margin, h - 1 - margin, rows)
    pts = []
    for j in range(rows):
        for i in range(cols):
            if len(pts) < n:
                jx = int(rng.integers(-3, 4))
                jy = int(rng.integers(-3, 4))
                pts.append((int(np.clip(xs[i] + jx, margin, w - 1 - margin)),
                            int(np.clip(ys[j] + jy, margin, h - 1 - margin))))
    return pts


def cut_template(im, cx, cy, pw, ph):
    """Template whose centre pixel (pw//2, ph//2) sits on (cx, cy); must lie inside the image."""
    x0, y0 = cx - pw // 2, cy - ph // 2
    assert x0 >= 0 and y0 >= 0 and x0 + pw <= im.shape[1] and y0 + ph <= im.shape[0]
    return im[y0:y0 + ph, x0:x0 + pw].copy()


def ellipse_mask(aw, ah, a, b, theta):
    """0/1 mask of an ellipse (semi-axes a, b, angle theta) centred in an aw x ah window."""
    y, x = np.mgrid[0:ah, 0:aw]
    xc, yc = (aw - 1) / 2.0, (ah - 1) / 2.0
    c, s = np.cos(theta), np.sin(theta)
    u = (x - xc) * c + (y - yc) * s
    v = -(x - xc) * s + (y - yc) * c
    return ((u / max(a, 0.5)) ** 2 + (v / max(b, 0.5)) ** 2 <= 1.0).astype(np.uint8)


def camera(w, h, rng, tilt=0.05):
    """A 3x4 projection matrix P = K [R | -R t] (float32) of a camera looking down +z with a small random
    rotation and offset: the `state.P` that Feature::project multiplies with B (feature.cpp:62)."""
    f = float(rng.uniform(0.8, 1.2)) * w
    K = np.array([[f, 0, w / 2.0], [0, f, h / 2.0], [0, 0, 1.0]])
    ax, ay, az = rng.uniform(-tilt, tilt, 3)
    Rx = np.array([[1, 0, 0], [0, np.cos(ax), -np.sin(ax)], [0, np.sin(ax), np.cos(ax)]])
    Ry = np.array([[np.cos(ay), 0, np.sin(ay)], [0, 1, 0], [-np.sin(ay), 0, np.cos(ay)]])
    Rz = np.array([[np.cos(az), -np.sin(az), 0], [np.sin(az), np.cos(az), 0], [0, 0, 1]])
    R = Rz @ Ry @ Rx
    t = rng.uniform(-0.05, 0.05, 3)
    P = K @ np.hstack([R, (-R @ t)[:, None]])
    return P.astype(np.float32)


def plane_feature(P, cx, cy, rng, depth=(2.0, 6.0), slant=0.4, scale=(0.8, 1.25)):
    """A planar feature seen at image point (cx, cy): returns (B, p3d, p2d) with B the 4x3 matrix of
    SLAM::Feature (feature.cpp:29-33: two in-plane directions scaled to about one descriptor pixel per
    image pixel, last row 0 0 1; column 2 is filled with p3d by project) and p2d = P [p3d; 1]
    dehomogenised (SLAM.hpp:150-158)."""
    P = np.asarray(P, np.float64)
    M, c = P[:, :3], P[:, 3]
    z = float(rng.uniform(*depth))
    ray = np.linalg.solve(M, np.array([cx, cy, 1.0]))
    p3d = np.linalg.solve(M, np.array([cx, cy, 1.0]) * z - c)       # M p3d + c = z (cx, cy, 1)
    n = ray / np.linalg.norm(ray) + rng.uniform(-slant, slant, 3)
    n /= np.linalg.norm(n)
    u = np.cross(n, [0.0, 1.0, 0.0]); u /= np.linalg.norm(u)
    v = np.cross(n, u); v /= np.linalg.norm(v)
    if (M @ u)[0] < 0:
        u = -u
    if (M @ v)[1] < 0:
        v = -v
    pix = z / np.linalg.norm(M[0])                                   # scene length of one pixel at depth z
    B = np.zeros((4, 3))
    B[:3, 0] = u * pix * float(rng.uniform(*scale))
    B[:3, 1] = v * pix * float(rng.uniform(*scale))
    B[3, 2] = 1.0
    B = B.astype(np.float32)
    p3 = p3d.astype(np.float32)
    a = P.astype(np.float32).astype(np.float64) @ np.append(p3.astype(np.float64), 1.0)
    p2d = (np.float32(a[0] / a[2]), np.float32(a[1] / a[2]))
    return B, p3, p2d
