"""What bench.py, tools/roofline.py, tools/bench_extras.py and the profiling tools share: the workload's shape and the
synthetic frames (a camera panning over one seeded world canvas).  bench.py re-exports these names."""
import numpy as np

W, H = 1920, 1080
FRAMES_PER_STEP = 256
CAP = 8192              # descriptor block capacity per frame (cv-sfm tracking_features, settings.rs:433-434)


def make_world(seed, w, h):
    """Deterministic synthetic 'world' canvas (value noise + rectangles + discs), uint8, numpy."""
    rng = np.random.default_rng(seed)
    img = np.full((h, w), 96.0, np.float32)
    for cell, amp in ((64, 48), (32, 24), (16, 12), (8, 6)):
        gh, gw = h // cell + 2, w // cell + 2
        g = rng.uniform(-amp, amp, (gh, gw)).astype(np.float32)
        ys = np.arange(h, dtype=np.float32) / cell
        xs = np.arange(w, dtype=np.float32) / cell
        y0 = ys.astype(int); x0 = xs.astype(int)
        fy = (ys - y0)[:, None]; fx = (xs - x0)[None, :]
        img += ((1 - fy) * (1 - fx) * g[y0][:, x0] + (1 - fy) * fx * g[y0][:, x0 + 1]
                + fy * (1 - fx) * g[y0 + 1][:, x0] + fy * fx * g[y0 + 1][:, x0 + 1])
    density = (w * h) / (1920.0 * 1080.0)
    n_shapes = int(200 * density)
    for _ in range(n_shapes):
        sw, sh = rng.integers(8, 97, 2)
        x, y = rng.integers(0, w), rng.integers(0, h)
        img[y:y + sh, x:x + sw] = rng.integers(0, 256)
    yy, xx = np.mgrid[0:97, 0:97]
    for _ in range(n_shapes):
        r = int(rng.integers(4, 49))
        x, y = int(rng.integers(r, w - r)), int(rng.integers(r, h - r))
        m = (yy[:2 * r + 1, :2 * r + 1] - r) ** 2 + (xx[:2 * r + 1, :2 * r + 1] - r) ** 2 <= r * r
        img[y - r:y + r + 1, x - r:x + r + 1][m] = rng.integers(0, 256)
    return np.clip(img, 0, 255).astype(np.uint8)


def make_frames(torch, device, rank, n_frames, world_size):
    """n_frames 1080p frames for this rank: a camera panning over the world canvas (4 px right, 2 px down
    per GLOBAL frame) plus +-2 sensor noise.  Global frame g = j*world_size + rank."""
    total = n_frames * world_size
    world = make_world(0xA4A2E, W + 4 * total + 64, H + 2 * total + 64)
    wt = torch.from_numpy(world).to(device)
    frames = torch.empty((n_frames, H, W), dtype=torch.uint8, device=device)
    gen = torch.Generator(device=device)
    for j in range(n_frames):
        g = j * world_size + rank
        gen.manual_seed(1000 + g)
        crop = wt[2 * g:2 * g + H, 4 * g:4 * g + W].to(torch.int16)
        noise = torch.randint(-2, 3, (H, W), generator=gen, device=device, dtype=torch.int16)
        frames[j] = (crop + noise).clamp_(0, 255).to(torch.uint8)
    return frames
