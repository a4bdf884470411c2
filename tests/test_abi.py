"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports every
symbol include/akz.h declares; the host-side schedule math matches the oracle.  No compute calls."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from cv_amd import build, _lib
    build.build()
    return _lib.lib()


def test_library_exports_every_declared_symbol(lib):
    from cv_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "akz.h")).read()
    declared = set(re.findall(r"\b((?:akz|hm|rs)_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"akz_status"}
    assert declared, "no declarations parsed"
    assert declared == set(_lib.ABI_SYMBOLS), declared ^ set(_lib.ABI_SYMBOLS)
    for name in sorted(declared):
        assert hasattr(lib, name), f"libakz.so does not export {name}"


def test_abi_number_is_the_headers(lib):
    """akz_abi_version() of the built library == AKZ_ABI_VERSION of include/akz.h == what every binding was written against
    (cv_amd/_lib.py refuses to load anything else; include/akaze.hpp and rust/akaze-mi355x check the same number)."""
    from cv_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "akz.h")).read()
    n = int(re.search(r"#define\s+AKZ_ABI_VERSION\s+(\d+)u", hdr).group(1))
    assert lib.akz_abi_version() == n == _lib.ABI_VERSION
    rs = open(os.path.join(ROOT, "rust", "akaze-mi355x", "src", "lib.rs")).read()
    assert int(re.search(r"const AKZ_ABI_VERSION: u32 = (\d+);", rs).group(1)) == n
    assert "require_abi()" in open(os.path.join(ROOT, "include", "akaze.hpp")).read()
    assert lib.hm_targets_generation(None) == 0


def test_pod_layouts_match_the_reference_types(lib):
    from cv_amd import _lib
    assert C.sizeof(_lib.Config) == 80
    cfg = _lib.Config()
    lib.akz_config_default(C.byref(cfg))
    # Akaze::default(), akaze/src/lib.rs:169-185
    assert cfg.maximum_features == 2 ** 64 - 1 and cfg.num_sublevels == 4 and cfg.max_octave_evolution == 4
    assert cfg.base_scale_offset == 1.6 and cfg.initial_contrast == 0.001 and cfg.contrast_percentile == 0.7
    assert cfg.contrast_factor_num_bins == 300 and cfg.derivative_factor == 1.5
    assert cfg.detector_threshold == 0.001 and cfg.descriptor_channels == 3 and cfg.descriptor_pattern_size == 10
    assert _lib.KP_DTYPE.itemsize == 28


def test_error_strings_and_no_cpu_fallback(lib):
    from cv_amd import _lib
    assert lib.akz_strerror(0) == b"ok"
    for code in range(-7, 0):
        assert lib.akz_strerror(code) not in (b"ok", b"unknown status")
    import torch
    if not torch.cuda.is_available():
        cfg = _lib.Config()
        lib.akz_config_default(C.byref(cfg))
        h = C.c_void_p()
        assert lib.akz_create(C.byref(cfg), 0, 64, 64, 1, 0, C.byref(h)) == -2  # AKZ_E_NO_DEVICE, no fallback
        hm = C.c_void_p()
        assert lib.hm_create(0, 16, 16, C.byref(hm)) == -2
        from cv_amd.akaze import Akaze
        with pytest.raises(_lib.AkzError):
            Akaze.sparse().extract(np.zeros((64, 64), np.uint8))


def test_options_struct_and_early_validation(lib):
    """akz_options is 64 bytes; akz_create_ex validates the configuration and the options BEFORE it looks for a
    device, so the refusals are checkable without a GPU: a pyramid of more than 32 levels (every per-frame level
    table has 32 slots — the check sits in front of every launch), malformed options."""
    from cv_amd import _lib
    assert C.sizeof(_lib.Options) == 64
    cfg = _lib.Config()
    lib.akz_config_default(C.byref(cfg))
    h = C.c_void_p()
    cfg.num_sublevels, cfg.max_octave_evolution = 8, 5          # 40 levels at 1920x1080
    assert lib.akz_create_ex(C.byref(cfg), 0, 1920, 1080, 1, 0, None, C.byref(h)) == -1
    cfg.num_sublevels, cfg.max_octave_evolution = 9, 4
    assert lib.akz_create_ex(C.byref(cfg), 0, 640, 480, 1, 0, None, C.byref(h)) == -1
    lib.akz_config_default(C.byref(cfg))
    o = _lib.make_options()
    o.reserved[3] = 1
    assert lib.akz_create_ex(C.byref(cfg), 0, 640, 480, 1, 0, C.byref(o), C.byref(h)) == -1
    o = _lib.make_options(fed_block=9)
    assert lib.akz_create_ex(C.byref(cfg), 0, 640, 480, 1, 0, C.byref(o), C.byref(h)) == -1
    o = _lib.make_options(arith=8)                              # three switches: 0..7
    assert lib.akz_create_ex(C.byref(cfg), 0, 640, 480, 1, 0, C.byref(o), C.byref(h)) == -1
    o = _lib.make_options()
    o.struct_size = 4
    assert lib.akz_create_ex(C.byref(cfg), 0, 640, 480, 1, 0, C.byref(o), C.byref(h)) == -1
    # a frame the 32-bit byte offsets of the diffusion / determinant kernels cannot address (akz_common.h kAkzMaxPixels)
    lib.akz_config_default(C.byref(cfg))
    assert lib.akz_create_ex(C.byref(cfg), 0, 32768, 16384, 1, 0, None, C.byref(h)) == -6        # AKZ_E_TOO_LARGE
    assert lib.akz_create_ex(C.byref(cfg), 0, 16385, 16384, 1, 0, None, C.byref(h)) == -6
    assert lib.akz_create(C.byref(cfg), 0, 65535, 65535, 1, 0, C.byref(h)) == -6
    # a valid call gets as far as the device probe
    import torch
    if not torch.cuda.is_available():
        assert lib.akz_create_ex(C.byref(cfg), 0, 640, 480, 1, 0, C.byref(_lib.make_options(keep_all=True)), C.byref(h)) == -2


def test_library_reads_no_environment_variable():
    """Behaviour switches live in akz_options / hm_create_ex flags: no getenv in the library sources."""
    import glob
    for f in glob.glob(os.path.join(ROOT, "cv_amd", "csrc", "*")):
        assert "getenv" not in open(f).read(), f


def test_host_mirror_surface():
    """The Python mirror keeps the reference's names (akaze/src/lib.rs:109-185, 295-366)."""
    from cv_amd.akaze import Akaze, KeyPoint
    a = Akaze.default()
    assert a.detector_threshold == 0.001 and Akaze.sparse().detector_threshold == 0.01
    assert Akaze.dense().detector_threshold == 0.0001 and Akaze.new(0.5).detector_threshold == 0.5
    for name in ("extract", "extract_from_gray_float_image", "extract_path"):
        assert callable(getattr(a, name))
    kp = KeyPoint((1.0, 2.0), 0.5, 4.8, 0, 0, 0.0)
    assert kp.image_point() == (1.0, 2.0)
    from cv_amd import knn
    for name in ("LinearKnn", "Hamming", "matching", "symmetric_matching", "match_descriptors"):
        assert hasattr(knn, name)


def test_gaussian_kernel_host_entry(lib):
    """akz_gaussian_kernel is host-only scalar math (image.rs:360-374): usable without a GPU."""
    out = np.empty(7, np.float32)
    assert lib.akz_gaussian_kernel(3.0, 7, out.ctypes.data) == 0
    known = [0.10628852, 0.14032133, 0.16577007, 0.17524014, 0.16577007, 0.14032133, 0.10628852]
    assert np.all(np.abs(out - np.array(known, np.float32)) < 1e-4)
    assert lib.akz_gaussian_kernel(3.0, 6, out.ctypes.data) == -1  # even size: the reference asserts


def test_committed_bench_line_follows_the_contract():
    """profiles/r05_bench.json is the JSON line bench.py printed on the MI355X: the driver's keys, the roofline object
    of the run's dominant kernel family (its own roofline fraction, so <= 1), the same for the top five, the parity
    check against the oracle, the extra BASELINE configs and the bounded CPU baseline must all be there."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "profiles", "r05_bench.json")) as f:
        d = json.loads(f.read().strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "roofline_top", "cpu_baseline", "parity_checked",
              "configs_extra"):
        assert k in d, k
    assert d["unit"] == "frames/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
    for r in [d["roofline"]] + d["roofline_top"]:
        for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "frac_is", "rank_ms_per_step"):
            assert k in r, k
        # frac is ALWAYS the family's own roofline fraction (bytes / 8 TB/s; MFMA ops / 10 PF for the matcher) — the VALU
        # issue fraction sits beside it as valu_frac and `bound` names the larger of the two
        assert r["bound"] in ("hbm", "valu", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0 < r["frac"] <= 1.0
        assert (r["unit"], r["peak"]) in (("GB/s", 8000.0), ("TOP/s", 10000.0))
        if r["unit"] == "GB/s":
            assert r["frac"] == r["hbm_frac"]
            if r["bound"] == "valu":
                assert r["valu_frac"] > r["hbm_frac"]
        if r["traffic"] is not None:       # counters were taken at this micro-batch; the kernel moves >= its algorithmic bytes
            assert r["traffic_source"]["micro_batch"] == d["config"]["micro_batch"]
            assert r["traffic"] >= 0.98 * r["algorithmic_bytes_per_launch"]
    assert d["roofline"]["kernel"] == d["roofline_top"][0]["kernel"]
    ranks = [r["rank_ms_per_step"] for r in d["roofline_top"]]
    assert ranks == sorted(ranks, reverse=True)                 # ordered by time per step with the GPU to itself
    assert any(r["bound"] == "mfma" for r in d["roofline_top"])  # the matcher is one of the families
    assert d["parity_checked"]["mismatches"] == 0 and d["parity_checked"]["frames"] >= 2
    for cfg in ("configs[2]", "configs[3]"):
        assert d["configs_extra"][cfg]["parity"]["mismatches"] == 0 and "roofline" in d["configs_extra"][cfg]
    assert 0 < d["configs_extra"]["configs[3]"]["roofline"]["frac"] <= 1.0            # FP64 issue fraction from committed counters
    reg = d["configs_extra"]["pipeline+register"]
    assert reg["parity"]["mismatches"] == 0 and reg["registered_frames_per_s"] > 500 and reg["frames_with_a_model"] == d["config"]["frames_per_gpu_per_step"]
    crit = d["configs_extra"]["criterion"]
    assert crit["mismatches"] == 0 and set(crit["rows"]) >= {"extract", "horizontal_filter_small_kernel", "vertical_filter_large_kernel"}
    assert d["cpu_baseline_intra_frame"]["value"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert d["value"] > 2000.0          # BASELINE.json's target for one MI355X


def test_headline_is_compact_and_parseable():
    """The driver keeps a bounded tail of stdout (the 22 KB line of round 4 left BENCH_r04.parsed null): bench.py's LAST
    stdout line is the headline built by bench.headline() — under 4 KB, with the contract keys, one flat roofline object,
    one cpu_baseline object, the parity counts and one scalar per extra leg; the full report goes to a side file."""
    import json
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    with open(os.path.join(root, "profiles", "r05_bench.json")) as f:
        full = json.loads(f.read().strip().splitlines()[-1])
    line = bench.headline(full, "gpurun_out/bench_detail.json")
    assert len(line) < 4096 and "\n" not in line
    h = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity", "extras"):
        assert k in h, k
    assert h["value"] == full["value"] and h["ms_per_step"] == full["ms_per_step"]
    r = h["roofline"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us", "launches"):
        assert k in r, k
    assert all(not isinstance(v, (dict, list)) for v in r.values())          # one flat object, no nested models
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0 < r["frac"] <= 1
    assert h["cpu_baseline"]["value"] > 0 and h["cpu_baseline"]["kind"] in ("port", "reference") and h["cpu_baseline"]["cores"] >= 1
    assert h["parity"]["mismatches"] == 0 and h["parity"]["frames"] >= 2
    assert set(h["extras"]) >= {"configs[2]", "configs[3]", "pipeline+verify", "pipeline+register"}
    # the line the run itself printed (what the driver parses) is the same construction
    with open(os.path.join(root, "profiles", "r05_bench_headline.json")) as f:
        printed = f.read().strip()
    assert len(printed) < 4096 and json.loads(printed)["value"] == full["value"] and json.loads(printed)["roofline"]["frac"] == r["frac"]
    # a report stuffed far beyond anything real still yields a line under the limit (optional objects are shed first)
    fat = dict(full)
    fat["config"] = dict(full["config"], workload="x" * 6000)
    assert len(bench.headline(fat, None)) < 4096


def test_gpus_n_without_a_launcher_starts_its_own_ranks(monkeypatch):
    """`python bench.py --gpus 2 ...` with no WORLD_SIZE in the environment (the driver's N = 1 command shape with another
    N) must become the launcher of its own ranks: exec of torch.distributed.run on 127.0.0.1 with the same arguments."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    seen = {}

    def fake_execve(exe, argv, env):
        seen["exe"], seen["argv"], seen["env"] = exe, list(argv), dict(env)
        raise SystemExit(0)
    import torch
    monkeypatch.setattr(os, "execve", fake_execve)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "20", "--warmup", "5"])
    # a node with fewer GPUs than ranks: a clear refusal instead of N ranks failing one by one
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    try:
        bench.main()
        assert False, "expected a refusal"
    except SystemExit as e:
        assert "--gpus 2" in str(e.code) and "seen" not in seen
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    try:
        bench.main()
    except SystemExit as e:
        assert e.code == 0
    a = seen["argv"]
    assert seen["exe"] == sys.executable and a[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node=2" in a and a[a.index("--master-addr") + 1] == "127.0.0.1"
    assert a[-6:] == ["--gpus", "2", "--steps", "20", "--warmup", "5"] and a[-7].endswith("bench.py")
    assert seen["env"].get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"


def test_pixel_normalisation_shortcut_is_exact():
    """cv_amd/csrc/akz_scale_space.hip: px_over_255 / px_over_65535 replace the reference's per-pixel IEEE division
    (image.rs:54, :57-66) by q0 = v * r, q = fma(fma(-d, q0, v), r, q0).  Exhaustive check over every u8 and u16 value
    (the fma is emulated in f64: every product and sum here is exact in 53 bits, so one rounding to f32 remains)."""
    import numpy as np

    def fma(a, b, c):
        return np.float32(np.float64(a) * np.float64(b) + np.float64(c))

    for d_, n in ((255.0, 256), (65535.0, 65536)):
        d = np.float32(d_)
        r = np.float32(np.float32(1.0) / d)
        v = np.arange(n, dtype=np.float32)
        want = (v / d).astype(np.float32)
        q0 = (v * r).astype(np.float32)
        got = np.array([fma(fma(-d, q0[i], v[i]), r, q0[i]) for i in range(n)], np.float32)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        assert not np.array_equal(q0.view(np.uint32), want.view(np.uint32))      # the multiply alone is NOT enough


def test_profiled_line_agrees_with_the_committed_rocprof_statistics():
    """profiles/r05_bench_profiled.json is the line `bench.py --steps 30 ...` printed UNDER rocprofv3 --kernel-trace, and
    profiles/r05_bench_kernel_stats.txt the per-kernel statistics of that very trace (tools/refresh_profiles.sh).  A family's
    `avg_launch_us` in the line (the launches' own start / stop events over the 30 timed steps) and the trace's average
    duration of the same kernel (all 34 launches of the process: warm-up and the instrumented pass included) are the same
    measurement taken twice: they must agree to within a few per cent, family by family."""
    import json
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "profiles", "r05_bench_profiled.json")) as f:
        d = json.loads(f.read().strip().splitlines()[-1])
    stats = {}
    with open(os.path.join(root, "profiles", "r05_bench_kernel_stats.txt")) as f:
        for line in f.read().splitlines()[1:]:
            m = re.match(r"(.+?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", line)
            if m:
                stats[m.group(1).strip()] = (int(m.group(2)), float(m.group(4)))
    # (the family k_front_fed<s,..> is the first octave's kernel, one template instance; the same kernel below the first
    # octave is several instances — halo depth, ring, Lflow written or not — under one family k_front_fed<s,2,..>: skipped)
    trace_name = {"k_orient_describe": "k_orient_describe", "k_front_fed<3,..>": "k_front_fed<3, 1, false, false>",
                  "k_front_fed<4,..>": "k_front_fed<4, 1, true, false>", "k_det_stream<3": "k_det_stream<3, false>",
                  "k_det_stream<4": "k_det_stream<4, false>", "k_det_stream<2": "k_det_stream<2, false>",
                  "k_level_front2<4,2,..,u8>": "k_level_front2<4, 2, 32, 512, unsigned char, false, 1>",
                  "k_knn_mfma4w<2>": "k_knn_mfma4w<2>"}
    seen = 0
    for r in d["roofline_top"]:
        key = next((k for k in trace_name if r["kernel"].startswith(k)), None)
        if key is None:
            assert r["kernel"].startswith(("k_front_fed<2,2", "k_front_fed<3,2", "k_front_fed<4,2", "k_contrast_pair")), r["kernel"]
            continue
        calls, avg_us = stats[trace_name[key]]
        assert calls >= r["launches"]
        assert abs(r["avg_launch_us"] - avg_us) <= 0.08 * avg_us, (r["kernel"], r["avg_launch_us"], avg_us)
        seen += 1
    assert seen >= 4


def test_compiled_form_of_the_kernels_the_compiler_can_ruin():
    """tools/check_isa.py on the built objects: k_rsb_score_p3p keeps the exact statement of its inlier test behind a branch and
    its match loop unrolled (the same source has compiled to forms 1.3 and 3.7 times slower)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from cv_amd import build
    build.build()
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_isa.py")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr


def test_experiment_patches_still_apply():
    """tools/variants/*.patch put experiment knobs, profilers and measured-and-rejected kernels back into the shipped sources
    (docs/EXPERIMENTS.md cites them): each must still apply to the tree it sits in."""
    import glob
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if shutil.which("git") is None:
        pytest.skip("no git")
    patches = sorted(glob.glob(os.path.join(root, "tools", "variants", "*.patch")))
    assert len(patches) >= 5
    for p in patches:
        r = subprocess.run(["git", "apply", "--check", p], cwd=root, capture_output=True, text=True)
        assert r.returncode == 0, (os.path.basename(p), r.stderr[-400:])
