//! `akaze`-compatible front-end over the MI355X library (C ABI declared in `include/akz.h`).
//!
//! NOT BUILT in this repository (no Rust toolchain in the build image); kept as the reference-side
//! binding of INTEGRATION.md.  Its C++ twin, include/akaze.hpp, has the same structure (per-thread context cache, capacity
//! growth, consensus layer) and IS built and run against the GPU by tests/cpp/estimate_pose.cpp.  The public surface is the one rust-cv callers use
//! (`akaze/src/lib.rs:69-185,295-366` of rust-cv/cv): `Akaze` with its 11 public fields,
//! `Akaze::{new, sparse, dense, extract, extract_from_gray_float_image, extract_path}`, `KeyPoint`, the `image`
//! module (`GrayFloatImage`, the separable filters, `gaussian_kernel`), a `space::Knn` implementor, the two-view
//! consensus and the place-recognition hasher.
use bitarray::BitArray;
use cv_core::{
    nalgebra::{IsometryMatrix3, Matrix3, Point2, Rotation3, Translation3, UnitVector3, Vector3},
    sample_consensus::{Consensus, Estimator},
    CameraToCamera, FeatureMatch, FeatureWorldMatch, ImagePoint, Projective, WorldToCamera,
};
use ::image::{DynamicImage, ImageResult};
use std::{cell::{Cell, RefCell}, os::raw::c_void, path::Path, ptr};

#[repr(C)]
#[derive(Clone, Copy)]
struct AkzConfig {
    maximum_features: u64,
    num_sublevels: u32,
    max_octave_evolution: u32,
    base_scale_offset: f64,
    initial_contrast: f64,
    contrast_percentile: f64,
    contrast_factor_num_bins: u64,
    derivative_factor: f64,
    detector_threshold: f64,
    descriptor_channels: u64,
    descriptor_pattern_size: u64,
}
#[repr(C)]
#[derive(Clone, Copy, Default)]
struct AkzKeypoint {
    x: f32,
    y: f32,
    response: f32,
    size: f32,
    angle: f32,
    octave: u32,
    class_id: u32,
}
#[repr(C)]
#[derive(Clone, Copy)]
pub struct AkzNeighbor {
    pub index: u32,
    pub distance: u32,
}

/// `akz_options` (include/akz.h): sixteen 32-bit words; everything zero = the library's defaults.
#[repr(C)]
#[derive(Clone, Copy, Default)]
struct AkzOptions {
    struct_size: u32,
    flags: u32,
    fed_block: u32,
    sup_capacity: u32,
    max_candidates: u32,
    desc_tile_shift: u32,
    stream_waves: u32,
    stream_min_waves: u32,
    arith: u32,
    cu_ss: u32,
    cu_kp: u32,
    resident_min_frames: u32,
    reserved: [u32; 4],
}

extern "C" {
    fn akz_create(cfg: *const AkzConfig, device: i32, max_w: i32, max_h: i32, max_batch: i32, max_kp: u32,
                  out: *mut *mut c_void) -> i32;
    fn akz_create_ex(cfg: *const AkzConfig, device: i32, max_w: i32, max_h: i32, max_batch: i32, max_kp: u32,
                     opts: *const AkzOptions, out: *mut *mut c_void) -> i32;
    fn akz_destroy(ctx: *mut c_void) -> i32;
    fn akz_extract_gray_u16(ctx: *mut c_void, img: *const u16, w: i32, h: i32, stride: i32, kps: *mut AkzKeypoint,
                            descs: *mut [u8; 64], cap: u32, n_out: *mut u32) -> i32;
    fn akz_extract_gray_u8(ctx: *mut c_void, img: *const u8, w: i32, h: i32, stride: i32, kps: *mut AkzKeypoint,
                           descs: *mut [u8; 64], cap: u32, n_out: *mut u32) -> i32;
    fn akz_extract_gray_f32(ctx: *mut c_void, img: *const f32, w: i32, h: i32, stride: i32, kps: *mut AkzKeypoint,
                            descs: *mut [u8; 64], cap: u32, n_out: *mut u32) -> i32;
    fn akz_extract_color(ctx: *mut c_void, pixels: *const c_void, fmt: i32, channels: i32, w: i32, h: i32, stride: i32,
                         kps: *mut AkzKeypoint, descs: *mut [u8; 64], cap: u32, n_out: *mut u32) -> i32;
    fn akz_gaussian_kernel(r: f32, kernel_size: u32, out: *mut f32) -> i32;
    fn akz_horizontal_filter(ctx: *mut c_void, img: *const f32, w: i32, h: i32, kernel: *const f32, ksize: u32,
                             out: *mut f32) -> i32;
    fn akz_vertical_filter(ctx: *mut c_void, img: *const f32, w: i32, h: i32, kernel: *const f32, ksize: u32,
                           out: *mut f32) -> i32;
    fn akz_half_size(ctx: *mut c_void, img: *const f32, w: i32, h: i32, out: *mut f32) -> i32;
    fn rs_create(device: i32, max_matches: u32, max_hypotheses: u32, out: *mut *mut c_void) -> i32;
    fn rs_destroy(ctx: *mut c_void) -> i32;
    fn rs_essential_batch(ctx: *mut c_void, bearings_a: *const f64, bearings_b: *const f64, n: u32,
                          sample_idx: *const u32, n_hyp: u32, thresh: f64, best_pose: *mut f64, best_id: *mut u32,
                          inlier_idx: *mut u32, cap: u32, n_inliers: *mut u32) -> i32;
    fn rs_essential_arrsac(ctx: *mut c_void, bearings_a: *const f64, bearings_b: *const f64, n: u32, sample_idx: *const u32,
                           params: *const RsArrsacParams, best_pose: *mut f64, best_id: *mut u32, inlier_idx: *mut u32,
                           cap: u32, n_inliers: *mut u32, stats: *mut c_void) -> i32;
    fn rs_p3p_arrsac(ctx: *mut c_void, bearings: *const f64, world: *const f64, n: u32, sample_idx: *const u32,
                     params: *const RsArrsacParams, best_pose: *mut f64, best_id: *mut u32, inlier_idx: *mut u32, cap: u32,
                     n_inliers: *mut u32, stats: *mut c_void) -> i32;
    fn hm_create(device: i32, max_q: u32, max_t: u32, out: *mut *mut c_void) -> i32;
    fn hm_destroy(ctx: *mut c_void) -> i32;
    fn hm_knn2(ctx: *mut c_void, q: *const [u8; 64], nq: u32, t: *const [u8; 64], nt: u32, out: *mut AkzNeighbor) -> i32;
    fn hm_knn(ctx: *mut c_void, q: *const [u8; 64], nq: u32, t: *const [u8; 64], nt: u32, k: u32,
              out: *mut AkzNeighbor) -> i32;
    fn hm_set_targets(ctx: *mut c_void, t: *const [u8; 64], nt: u32) -> i32;
    fn hm_targets_generation(ctx: *mut c_void) -> u64;
    fn akz_abi_version() -> u32;
    fn hm_knn_targets(ctx: *mut c_void, q: *const [u8; 64], nq: u32, k: u32, out: *mut AkzNeighbor) -> i32;
    fn hm_hash_bag(ctx: *mut c_void, feats: *const [u8; 64], n: u32, codewords: *const [u8; 64], n_codewords: u32,
                   hash: *mut u8, words: *mut AkzNeighbor) -> i32;
    fn hm_hash_knn(ctx: *mut c_void, query: *const u8, hashes: *const u8, n: u32, hash_bytes: u32, k: u32,
                   out: *mut AkzNeighbor, n_out: *mut u32) -> i32;
}

/// `akaze::KeyPoint` (akaze/src/lib.rs:69-93).
#[derive(Debug, Clone, Copy)]
pub struct KeyPoint {
    pub point: (f32, f32),
    pub response: f32,
    pub size: f32,
    pub octave: usize,
    pub class_id: usize,
    pub angle: f32,
}
impl ImagePoint for KeyPoint {
    fn image_point(&self) -> Point2<f64> {
        Point2::new(self.point.0 as f64, self.point.1 as f64)
    }
}

/// `akaze::Akaze` (akaze/src/lib.rs:109-185): same fields, same defaults.
#[derive(Debug, Copy, Clone)]
pub struct Akaze {
    pub maximum_features: usize,
    pub num_sublevels: u32,
    pub max_octave_evolution: u32,
    pub base_scale_offset: f64,
    pub initial_contrast: f64,
    pub contrast_percentile: f64,
    pub contrast_factor_num_bins: usize,
    pub derivative_factor: f64,
    pub detector_threshold: f64,
    pub descriptor_channels: usize,
    pub descriptor_pattern_size: usize,
}
impl Default for Akaze {
    fn default() -> Akaze {
        Akaze {
            maximum_features: usize::MAX,
            num_sublevels: 4,
            max_octave_evolution: 4,
            base_scale_offset: 1.6,
            initial_contrast: 0.001,
            contrast_percentile: 0.7,
            contrast_factor_num_bins: 300,
            derivative_factor: 1.5,
            detector_threshold: 0.001,
            descriptor_channels: 3,
            descriptor_pattern_size: 10,
        }
    }
}

/// `rs_arrsac_params` (include/akz.h).
#[repr(C)]
#[derive(Clone, Copy)]
pub struct RsArrsacParams {
    struct_size: u32,
    n_hypotheses: u32,
    block_size: u32,
    init_blocks: u32,
    max_candidates: u32,
    flags: u32,
    threshold: f64,
    sprt_delta: f64,
    sprt_ratio: f64,
    seed: u64,
    estimations_per_block: u32,
    reserved: u32,
}

const AKZ_E_CAPACITY: i32 = -4;
const AKZ_E_INTERNAL: i32 = -7;
const FMT_U8: i32 = 0;
const FMT_F32: i32 = 1;
const FMT_U16: i32 = 2;
/// First capacity of the device's per-frame lists.  The reference's `Vec`s are unbounded (`maximum_features` is
/// `usize::MAX`, lib.rs:172): a call that overflows the capacity is repeated with twice as much, up to the library's
/// 262 144 keypoints per frame — a caller never sees the cap, only (beyond 262 144) a panic naming it.
const INITIAL_KP: u32 = 16384;
const LIBRARY_MAX_KP: u32 = 262144;

/// One device pyramid per (config, size, capacity), per thread.  `Akaze` itself stays the reference's `Copy` struct
/// with `&self` methods and no state (lib.rs:108): a loop of `extract` calls (cv-sfm/src/lib.rs:2200-2204) re-uses the
/// pyramid instead of allocating ~0.2 GB per frame.
struct CachedCtx {
    cfg: AkzConfig,
    w: u32,
    h: u32,
    cap: u32,
    arith: u32,
    ctx: *mut c_void,
}
impl Drop for CachedCtx {
    fn drop(&mut self) {
        unsafe { akz_destroy(self.ctx) };
    }
}
thread_local! {
    static CTX: RefCell<Option<CachedCtx>> = RefCell::new(None);
    static MATCHER: RefCell<Option<(u32, *mut c_void)>> = RefCell::new(None);
}
fn same_cfg(a: &AkzConfig, b: &AkzConfig) -> bool {
    a.maximum_features == b.maximum_features && a.num_sublevels == b.num_sublevels
        && a.max_octave_evolution == b.max_octave_evolution && a.base_scale_offset == b.base_scale_offset
        && a.initial_contrast == b.initial_contrast && a.contrast_percentile == b.contrast_percentile
        && a.contrast_factor_num_bins == b.contrast_factor_num_bins && a.derivative_factor == b.derivative_factor
        && a.detector_threshold == b.detector_threshold && a.descriptor_channels == b.descriptor_channels
        && a.descriptor_pattern_size == b.descriptor_pattern_size
}
/// Which of the reference's three un-vendored arithmetic orders the filters and `half_size` use (`akz_options.arith`,
/// `AKZ_ARITH_*` bits of include/akz.h; 0 = what the crate sources imply for a default x86-64 build).  The `Akaze` struct
/// stays the reference's eleven fields, so this is process-wide: call it once, before the first `extract`, with the value
/// `python3 tools/pin_arith.py <rust>_kps.csv <rust>_descs.txt image.png` names for the cargo build this crate replaces
/// (INTEGRATION.md 6).  Contexts created under another value are dropped at the next call.
pub fn set_arith(arith: u32) {
    assert!(arith < 8, "akz_options.arith takes AKZ_ARITH_* bits 0..7");
    ARITH.store(arith, std::sync::atomic::Ordering::Relaxed);
}
static ARITH: std::sync::atomic::AtomicU32 = std::sync::atomic::AtomicU32::new(0);

/// The thread's matcher context, grown to hold `n` descriptors a side.
/// include/akz.h AKZ_ABI_VERSION these bindings were written against; the loaded library must export the same number.
const AKZ_ABI_VERSION: u32 = 8;
fn require_abi() {
    let got = unsafe { akz_abi_version() };
    assert_eq!(got, AKZ_ABI_VERSION, "libakz exports ABI {got}, akaze-mi355x was written against {AKZ_ABI_VERSION}");
}

fn with_matcher<R>(n: u32, f: impl FnOnce(*mut c_void) -> R) -> R {
    MATCHER.with(|m| {
        let mut m = m.borrow_mut();
        if m.map_or(true, |(cap, _)| cap < n) {
            if let Some((_, old)) = m.take() {
                unsafe { hm_destroy(old) };
            }
            require_abi();
            let cap = n.max(4096).next_power_of_two();
            let mut ctx: *mut c_void = ptr::null_mut();
            let st = unsafe { hm_create(0, cap, cap, &mut ctx) };
            assert_eq!(st, 0, "hm_create failed with status {st} (there is no CPU fallback)");
            *m = Some((cap, ctx));
        }
        f(m.unwrap().1)
    })
}

impl Akaze {
    pub fn new(threshold: f64) -> Self {
        Self { detector_threshold: threshold, ..Default::default() }
    }
    pub fn sparse() -> Self {
        Self::new(0.01)
    }
    pub fn dense() -> Self {
        Self::new(0.0001)
    }

    fn config(&self) -> AkzConfig {
        AkzConfig {
            maximum_features: self.maximum_features as u64,
            num_sublevels: self.num_sublevels,
            max_octave_evolution: self.max_octave_evolution,
            base_scale_offset: self.base_scale_offset,
            initial_contrast: self.initial_contrast,
            contrast_percentile: self.contrast_percentile,
            contrast_factor_num_bins: self.contrast_factor_num_bins as u64,
            derivative_factor: self.derivative_factor,
            detector_threshold: self.detector_threshold,
            descriptor_channels: self.descriptor_channels as u64,
            descriptor_pattern_size: self.descriptor_pattern_size as u64,
        }
    }

    /// One extraction through the thread's cached context, with growth: `AKZ_E_INTERNAL` (a device list overflowed) and
    /// `AKZ_E_CAPACITY` (more keypoints than the output arrays hold) both mean "not enough room"; the reference has no
    /// such condition, so the call is repeated with twice the capacity.
    fn run<F: Fn(*mut c_void, *mut AkzKeypoint, *mut [u8; 64], u32, *mut u32) -> i32>(
        &self, w: u32, h: u32, f: F,
    ) -> (Vec<KeyPoint>, Vec<BitArray<64>>) {
        let cfg = self.config();
        let mut cap = INITIAL_KP.min((self.maximum_features as u64).max(64).min(LIBRARY_MAX_KP as u64) as u32);
        loop {
            let (st, kps, descs, n) = CTX.with(|slot| {
                let mut slot = slot.borrow_mut();
                let arith = ARITH.load(std::sync::atomic::Ordering::Relaxed);
                let fits = slot.as_ref().map_or(false, |c| same_cfg(&c.cfg, &cfg) && c.cap == cap && c.arith == arith && w <= c.w && h <= c.h);
                if !fits {
                    *slot = None; // drops (and destroys) the previous context first
                    let mut ctx: *mut c_void = ptr::null_mut();
                    let opts = AkzOptions { struct_size: std::mem::size_of::<AkzOptions>() as u32, arith, ..Default::default() };
                    let st = unsafe { akz_create_ex(&cfg, 0, w as i32, h as i32, 1, cap, if arith != 0 { &opts } else { ptr::null() }, &mut ctx) };
                    assert_eq!(st, 0, "akz_create_ex failed with status {st} (there is no CPU fallback)");
                    *slot = Some(CachedCtx { cfg, w, h, cap, arith, ctx });
                }
                let ctx = slot.as_ref().unwrap().ctx;
                let mut kps = vec![AkzKeypoint::default(); cap as usize];
                let mut descs = vec![[0u8; 64]; cap as usize];
                let mut n = 0u32;
                let st = f(ctx, kps.as_mut_ptr(), descs.as_mut_ptr(), cap, &mut n);
                (st, kps, descs, n)
            });
            if (st == AKZ_E_INTERNAL || st == AKZ_E_CAPACITY) && cap < LIBRARY_MAX_KP {
                cap = (cap * 2).min(LIBRARY_MAX_KP);
                continue;
            }
            assert_eq!(st, 0, "akz_extract failed with status {st} (capacity {cap} keypoints per frame)");
            let n = n as usize;
            let keypoints = kps[..n]
                .iter()
                .map(|k| KeyPoint {
                    point: (k.x, k.y),
                    response: k.response,
                    size: k.size,
                    octave: k.octave as usize,
                    class_id: k.class_id as usize,
                    angle: k.angle,
                })
                .collect();
            let descriptors = descs[..n].iter().map(|d| BitArray::new(*d)).collect();
            return (keypoints, descriptors);
        }
    }

    /// `Akaze::extract` (akaze/src/lib.rs:295).  Every arm of `GrayFloatImage::from_dynamic` (image.rs:45-109) runs on the
    /// device: the gray ones as they are, the colour ones through `akz_extract_color` (`DynamicImage::grayscale()` there).
    pub fn extract(&self, image: &DynamicImage) -> (Vec<KeyPoint>, Vec<BitArray<64>>) {
        let (w, h) = (image.width(), image.height());
        let colour = |pixels: *const c_void, fmt: i32, channels: i32| {
            self.run(w, h, move |ctx, k, d, cap, n| unsafe {
                akz_extract_color(ctx, pixels, fmt, channels, w as i32, h as i32, w as i32 * channels, k, d, cap, n)
            })
        };
        match image {
            DynamicImage::ImageLuma8(g) => self.run(w, h, |ctx, k, d, cap, n| unsafe {
                akz_extract_gray_u8(ctx, g.as_raw().as_ptr(), w as i32, h as i32, w as i32, k, d, cap, n)
            }),
            DynamicImage::ImageLuma16(g) => self.run(w, h, |ctx, k, d, cap, n| unsafe {
                // image.rs:57-66: f32::from(v) / 65535f32, done on the device
                akz_extract_gray_u16(ctx, g.as_raw().as_ptr(), w as i32, h as i32, w as i32, k, d, cap, n)
            }),
            DynamicImage::ImageRgb8(c) => colour(c.as_raw().as_ptr() as *const c_void, FMT_U8, 3),
            DynamicImage::ImageRgba8(c) => colour(c.as_raw().as_ptr() as *const c_void, FMT_U8, 4),
            DynamicImage::ImageRgb16(c) => colour(c.as_raw().as_ptr() as *const c_void, FMT_U16, 3),
            DynamicImage::ImageRgba16(c) => colour(c.as_raw().as_ptr() as *const c_void, FMT_U16, 4),
            DynamicImage::ImageRgb32F(c) => colour(c.as_raw().as_ptr() as *const c_void, FMT_F32, 3),
            DynamicImage::ImageRgba32F(c) => colour(c.as_raw().as_ptr() as *const c_void, FMT_F32, 4),
            // LumaA8 / LumaA16 (image.rs:67-86: the luma sample of every pixel): drop the alpha on the host
            other => match other.grayscale() {
                DynamicImage::ImageLumaA8(g) => {
                    let l: Vec<u8> = g.pixels().map(|p| p[0]).collect();
                    self.run(w, h, |ctx, k, d, cap, n| unsafe {
                        akz_extract_gray_u8(ctx, l.as_ptr(), w as i32, h as i32, w as i32, k, d, cap, n)
                    })
                }
                DynamicImage::ImageLumaA16(g) => {
                    let l: Vec<u16> = g.pixels().map(|p| p[0]).collect();
                    self.run(w, h, |ctx, k, d, cap, n| unsafe {
                        akz_extract_gray_u16(ctx, l.as_ptr(), w as i32, h as i32, w as i32, k, d, cap, n)
                    })
                }
                _ => panic!("DynamicImage::grayscale() returned unexpected type"), // image.rs:107
            },
        }
    }

    /// `Akaze::extract_from_gray_float_image` (akaze/src/lib.rs:309).
    pub fn extract_from_gray_float_image(&self, img: &crate::image::GrayFloatImage) -> (Vec<KeyPoint>, Vec<BitArray<64>>) {
        let (w, h) = (img.width() as u32, img.height() as u32);
        self.run(w, h, |ctx, k, d, cap, n| unsafe {
            akz_extract_gray_f32(ctx, img.0.as_raw().as_ptr(), w as i32, h as i32, w as i32, k, d, cap, n)
        })
    }

    /// `Akaze::extract_path` (akaze/src/lib.rs:361).
    pub fn extract_path(&self, path: impl AsRef<Path>) -> ImageResult<(Vec<KeyPoint>, Vec<BitArray<64>>)> {
        Ok(self.extract(&::image::open(path)?))
    }
}

/// A `space::Knn` implementor with `LinearKnn { metric: Hamming, iter }` semantics for `BitArray<64>`.
///
/// The reference's callers build the `LinearKnn` once per frame pair and call `knn` once per query descriptor
/// (akaze/tests/estimate_pose.rs:82-88, cv-sfm/src/lib.rs:3103).  Ported literally that is one kernel launch per query:
/// the target set is uploaded ONCE (`hm_set_targets`: it stays on the device for as long as this thread keeps asking
/// about the same slice) and each `knn` sends 64 bytes up and `num` neighbours back — correct, and far better than
/// re-uploading 320 KB per query, but still launch-bound (about 20 us per query against 0.2 us per query for a whole
/// frame at once).  Callers that own their loop should ask for all queries together: [`Mi355xLinearKnn::knn_batch`], or
/// [`match_descriptors`] / `symmetric_matching`, which also keep the pair lists on the device side of the matcher.
pub struct Mi355xLinearKnn<'a> {
    pub targets: &'a [BitArray<64>],
    /// (matcher, generation) of THIS value's upload (`hm_targets_generation` right after its `hm_set_targets`).  The
    /// borrow keeps `targets` immutable for as long as the value lives, and the token lives in the value: a set dropped
    /// and another allocated at the same address is a different value with no token, so it uploads.
    upload: Cell<Option<(*mut c_void, u64)>>,
}
impl<'a> Mi355xLinearKnn<'a> {
    /// `LinearKnn { metric: Hamming, iter: targets }`.
    pub fn new(targets: &'a [BitArray<64>]) -> Self {
        Self { targets, upload: Cell::new(None) }
    }
    /// Upload the targets unless the matcher still holds exactly this value's upload: any other upload, or any
    /// host-buffer call that took the staging buffer, changes the matcher's generation number.
    fn resident(&self, ctx: *mut c_void) {
        let current = unsafe { hm_targets_generation(ctx) };
        if self.upload.get().map_or(true, |(c, g)| c != ctx || g != current || current == 0) {
            let st = unsafe { hm_set_targets(ctx, self.targets.as_ptr() as *const [u8; 64], self.targets.len() as u32) };
            assert_eq!(st, 0, "hm_set_targets failed with status {st}");
            self.upload.set(Some((ctx, unsafe { hm_targets_generation(ctx) })));
        }
    }
    /// `knn(q, num)` for every query in ONE launch: `out[i]` are the `min(num, targets.len())` nearest targets of
    /// `queries[i]`, ascending (distance, index) — what `queries.iter().map(|q| self.knn(q, num))` returns.
    pub fn knn_batch(&self, queries: &[BitArray<64>], num: usize) -> Vec<Vec<space::Neighbor<u32, usize>>> {
        assert!((1..=3).contains(&num), "the MI355X matcher implements knn(query, k) for k <= 3");
        let (nq, nt) = (queries.len() as u32, self.targets.len() as u32);
        let mut out = vec![AkzNeighbor { index: 0, distance: 0 }; queries.len() * num];
        let st = with_matcher(nq.max(nt), |ctx| {
            self.resident(ctx);
            unsafe { hm_knn_targets(ctx, queries.as_ptr() as *const [u8; 64], nq, num as u32, out.as_mut_ptr()) }
        });
        assert_eq!(st, 0);
        let keep = num.min(self.targets.len());
        out.chunks(num).map(|c| c.iter().take(keep)
            .map(|o| space::Neighbor { index: o.index as usize, distance: o.distance }).collect()).collect()
    }
}
impl<'a> space::Knn for Mi355xLinearKnn<'a> {
    type Ix = usize;
    type Metric = bitarray::Hamming;
    type Point = BitArray<64>;
    type KnnIter = Vec<space::Neighbor<u32, usize>>;
    fn knn(&self, query: &BitArray<64>, num: usize) -> Self::KnnIter {
        // cv-sfm asks for 2 when matching frame pairs (lib.rs:3103) and 3 when registering a frame (lib.rs:1474)
        self.knn_batch(std::slice::from_ref(query), num).pop().unwrap()
    }
    fn nn(&self, query: &BitArray<64>) -> Option<space::Neighbor<u32, usize>> {
        self.knn(query, 2).into_iter().next()
    }
}

/// `hamming_lsh::HammingHasher<64, 512>` as cv-sfm uses it (cv-sfm/src/lib.rs:205,216,672): a bag of descriptors
/// -> 512-byte hash over a 4096-word codebook, on the MI355X.  (hamming-lsh is not vendored in the reference; the
/// nearest-codeword form restated in oracle/lsh_oracle.c is what runs here.)
pub struct Mi355xHammingHasher {
    codewords: Vec<BitArray<64>>,
}
impl Mi355xHammingHasher {
    pub fn new_with_codewords(codewords: Vec<BitArray<64>>) -> Self {
        assert_eq!(codewords.len(), 512 * 8);
        Self { codewords }
    }
    pub fn hash_bag<'a>(&self, features: impl IntoIterator<Item = &'a BitArray<64>>) -> BitArray<512> {
        let feats: Vec<BitArray<64>> = features.into_iter().cloned().collect();
        let cap = (feats.len() as u32).max(self.codewords.len() as u32);
        let mut hash = [0u8; 512];
        let st = with_matcher(cap, |ctx| unsafe {
            hm_hash_bag(ctx, feats.as_ptr() as *const [u8; 64], feats.len() as u32,
                        self.codewords.as_ptr() as *const [u8; 64], self.codewords.len() as u32, hash.as_mut_ptr(),
                        ptr::null_mut())
        });
        assert_eq!(st, 0);
        BitArray::new(hash)
    }
}

/// `lsh_to_frame.knn_values(&lsh, num)` (cv-sfm/src/lib.rs:622-624) as an exact search over the stored hashes:
/// (index into `hashes`, distance), ascending (distance, index).
pub fn nearest_hashes(query: &BitArray<512>, hashes: &[BitArray<512>], num: usize) -> Vec<(usize, u32)> {
    let mut out = vec![AkzNeighbor { index: 0, distance: 0 }; num.max(1)];
    let mut n: u32 = 0;
    let st = with_matcher(2, |ctx| unsafe {
        hm_hash_knn(ctx, query.bytes().as_ptr(), hashes.as_ptr() as *const u8, hashes.len() as u32, 512, num as u32,
                    out.as_mut_ptr(), &mut n)
    });
    assert_eq!(st, 0);
    out.iter().take(n as usize).map(|o| (o.index as usize, o.distance)).collect()
}

/// `akaze::image` (akaze/src/image.rs): the f32 image wrapper and the separable filters, on the MI355X.
pub mod image {
    use super::*;
    use ::image::{ImageBuffer, Luma};

    pub type GrayImageBuffer = ImageBuffer<Luma<f32>, Vec<f32>>;

    /// `GrayFloatImage` (image.rs:36).
    #[derive(Debug, Clone)]
    pub struct GrayFloatImage(pub GrayImageBuffer);

    fn with_ctx<R>(w: u32, h: u32, f: impl FnOnce(*mut c_void) -> R) -> R {
        let cfg = Akaze::default().config();
        let mut ctx: *mut c_void = ptr::null_mut();
        assert_eq!(unsafe { akz_create(&cfg, 0, w as i32, h as i32, 1, 16, &mut ctx) }, 0);
        let r = f(ctx);
        unsafe { akz_destroy(ctx) };
        r
    }

    impl GrayFloatImage {
        /// image.rs:45-109 (8-bit gray takes v / 255 per pixel; everything else goes through `to_luma32f`)
        pub fn from_dynamic(input_image: &DynamicImage) -> Self {
            match input_image.grayscale() {
                DynamicImage::ImageLuma8(g) => Self(ImageBuffer::from_fn(g.width(), g.height(), |x, y| {
                    Luma([f32::from(g.get_pixel(x, y)[0]) / 255f32])
                })),
                other => Self(other.to_luma32f()),
            }
        }
        pub fn width(&self) -> usize { self.0.width() as usize }
        pub fn height(&self) -> usize { self.0.height() as usize }
        pub fn new(width: usize, height: usize) -> Self { Self(ImageBuffer::new(width as u32, height as u32)) }
        pub fn get(&self, x: usize, y: usize) -> f32 { self.0.get_pixel(x as u32, y as u32)[0] }
        pub fn put(&mut self, x: usize, y: usize, v: f32) { self.0.put_pixel(x as u32, y as u32, Luma([v])) }
        /// image.rs:154-199
        pub fn half_size(&self) -> Self {
            let (w, h) = (self.0.width(), self.0.height());
            let mut out = vec![0f32; ((w / 2) * (h / 2)) as usize];
            let st = with_ctx(w, h, |ctx| unsafe {
                akz_half_size(ctx, self.0.as_raw().as_ptr(), w as i32, h as i32, out.as_mut_ptr())
            });
            assert_eq!(st, 0);
            Self(ImageBuffer::from_raw(w / 2, h / 2, out).unwrap())
        }
    }

    fn filter(image: &GrayImageBuffer, kernel: &[f32], vertical: bool) -> GrayImageBuffer {
        let (w, h) = (image.width(), image.height());
        let mut out = vec![0f32; (w * h) as usize];
        let st = with_ctx(w, h, |ctx| unsafe {
            if vertical {
                akz_vertical_filter(ctx, image.as_raw().as_ptr(), w as i32, h as i32, kernel.as_ptr(), kernel.len() as u32,
                                    out.as_mut_ptr())
            } else {
                akz_horizontal_filter(ctx, image.as_raw().as_ptr(), w as i32, h as i32, kernel.as_ptr(),
                                      kernel.len() as u32, out.as_mut_ptr())
            }
        });
        assert_eq!(st, 0);
        ImageBuffer::from_raw(w, h, out).unwrap()
    }
    /// image.rs:202
    pub fn horizontal_filter(image: &GrayImageBuffer, kernel: &[f32]) -> GrayImageBuffer { filter(image, kernel, false) }
    /// image.rs:253
    pub fn vertical_filter(image: &GrayImageBuffer, kernel: &[f32]) -> GrayImageBuffer { filter(image, kernel, true) }
    /// image.rs:333
    pub fn separable_filter(image: &GrayImageBuffer, h_kernel: &[f32], v_kernel: &[f32]) -> GrayImageBuffer {
        vertical_filter(&horizontal_filter(image, h_kernel), v_kernel)
    }
    /// image.rs:360
    pub fn gaussian_kernel(r: f32, kernel_size: usize) -> Vec<f32> {
        let mut k = vec![0f32; kernel_size];
        assert_eq!(unsafe { akz_gaussian_kernel(r, kernel_size as u32, k.as_mut_ptr()) }, 0);
        k
    }
    /// image.rs:383
    pub fn gaussian_blur(image: &GrayFloatImage, r: f32) -> GrayFloatImage {
        assert!(r > 0.0, "sigma must be > 0.0");                       // image.rs:384
        let radius = (2.0 * r).ceil() as usize;                        // image.rs:385
        let k = gaussian_kernel(r, 2 * radius + 1);
        GrayFloatImage(separable_filter(&image.0, &k, &k))
    }
}

/// `arrsac::Arrsac` over the MI355X library: the `sample_consensus::Consensus` the reference's callers hand their
/// estimators to (akaze/tests/estimate_pose.rs:63-75, vslam-sandbox/src/main.rs:105-117, cv-sfm/src/lib.rs:1394-1412 and
/// 1619-1622), for the two estimators of the hot path — `EightPoint` over `FeatureMatch` and `LambdaTwist` over
/// `FeatureWorldMatch`.  The estimator argument selects the device procedure (`rs_essential_arrsac` / `rs_p3p_arrsac`);
/// its own `estimate` is not called: hypotheses, scoring and consensus all run on the GPU.  The `arrsac` crate is not
/// vendored in rust-cv/cv; what runs is this library's ARRSAC-shaped procedure (include/akz.h, specified by
/// oracle/arrsac_oracle.c) under the crate's builder names, seeded with a `u64` where the crate takes an `Rng`.
pub struct Arrsac {
    params: RsArrsacParams,
    ctx: Option<(u32, u32, *mut c_void)>, // (matches, hypotheses) the context holds
}
impl Arrsac {
    pub fn new(inlier_threshold: f64, seed: u64) -> Self {
        Self {
            params: RsArrsacParams {
                struct_size: std::mem::size_of::<RsArrsacParams>() as u32,
                // arrsac 0.10's documented defaults (the crate is not vendored in rust-cv/cv: unverified here):
                // initialization_hypotheses 256, block_size 100, initialization_blocks 4, max_candidate_hypotheses 50,
                // estimations_per_block 64, likelihood_ratio_threshold 1e3
                n_hypotheses: 256,
                block_size: 100,
                init_blocks: 4,
                max_candidates: 50,
                flags: 1 | 2 | 4, // RS_PRUNE_BOUND | RS_PRUNE_SPRT | RS_PRUNE_HALVE
                threshold: inlier_threshold,
                sprt_delta: 0.05,
                sprt_ratio: 1e3,
                seed,
                estimations_per_block: 64,
                reserved: 0,
            },
            ctx: None,
        }
    }
    pub fn initialization_hypotheses(mut self, n: usize) -> Self { self.params.n_hypotheses = n as u32; self }
    pub fn max_candidate_hypotheses(mut self, n: usize) -> Self { self.params.max_candidates = n as u32; self }
    pub fn estimations_per_block(mut self, n: usize) -> Self { self.params.estimations_per_block = n as u32; self }
    pub fn block_size(mut self, n: usize) -> Self { self.params.block_size = n as u32; self }
    pub fn initialization_blocks(mut self, n: usize) -> Self { self.params.init_blocks = n as u32; self }
    pub fn likelihood_ratio_threshold(mut self, r: f64) -> Self { self.params.sprt_ratio = r; self }

    fn context(&mut self, n: u32) -> *mut c_void {
        let blocks = (n + self.params.block_size - 1) / self.params.block_size;
        let need_h = self.params.n_hypotheses + self.params.estimations_per_block * blocks;
        if self.ctx.map_or(true, |(m, h, _)| m < n || h < need_h) {
            if let Some((_, _, old)) = self.ctx.take() {
                unsafe { rs_destroy(old) };
            }
            let mut ctx: *mut c_void = ptr::null_mut();
            let st = unsafe { rs_create(0, n.max(64), need_h, &mut ctx) };
            assert_eq!(st, 0, "rs_create failed with status {st} (there is no CPU fallback)");
            self.ctx = Some((n.max(64), need_h, ctx));
        }
        self.ctx.unwrap().2
    }
    /// (row-major 3x4 `[R | t]`, inlier indices) or `None`
    fn run(&mut self, p3p: bool, a: &[f64], b: &[f64], n: u32) -> Option<([f64; 12], Vec<usize>)> {
        let ctx = self.context(n);
        let mut pose = [0f64; 12];
        let (mut best, mut n_inl) = (0u32, 0u32);
        let mut inl = vec![0u32; n as usize];
        let st = unsafe {
            if p3p {
                rs_p3p_arrsac(ctx, a.as_ptr(), b.as_ptr(), n, ptr::null(), &self.params, pose.as_mut_ptr(), &mut best,
                              inl.as_mut_ptr(), n, &mut n_inl, ptr::null_mut())
            } else {
                rs_essential_arrsac(ctx, a.as_ptr(), b.as_ptr(), n, ptr::null(), &self.params, pose.as_mut_ptr(), &mut best,
                                    inl.as_mut_ptr(), n, &mut n_inl, ptr::null_mut())
            }
        };
        assert_eq!(st, 0, "consensus failed with status {st}");
        if best == u32::MAX {
            return None; // Consensus::model_inliers -> None: no sample produced a model
        }
        Some((pose, inl[..n_inl as usize].iter().map(|&i| i as usize).collect()))
    }
}
impl Drop for Arrsac {
    fn drop(&mut self) {
        if let Some((_, _, ctx)) = self.ctx.take() {
            unsafe { rs_destroy(ctx) };
        }
    }
}
fn isometry(rt: &[f64; 12]) -> IsometryMatrix3<f64> {
    let r = Matrix3::new(rt[0], rt[1], rt[2], rt[4], rt[5], rt[6], rt[8], rt[9], rt[10]);
    IsometryMatrix3::from_parts(Translation3::from(Vector3::new(rt[3], rt[7], rt[11])), Rotation3::from_matrix_unchecked(r))
}
/// `Consensus<EightPoint, FeatureMatch>` — eight-point/src/lib.rs:70-83 estimates, cv-core/src/pose.rs:249-295 scores.
impl Consensus<eight_point::EightPoint, FeatureMatch> for Arrsac {
    type Inliers = Vec<usize>;
    fn model<I>(&mut self, estimator: &eight_point::EightPoint, data: I) -> Option<CameraToCamera>
    where I: Iterator<Item = FeatureMatch> + Clone {
        self.model_inliers(estimator, data).map(|(m, _)| m)
    }
    fn model_inliers<I>(&mut self, _estimator: &eight_point::EightPoint, data: I) -> Option<(CameraToCamera, Vec<usize>)>
    where I: Iterator<Item = FeatureMatch> + Clone {
        let (mut a, mut b) = (Vec::new(), Vec::new());
        for FeatureMatch(fa, fb) in data {
            a.extend_from_slice(fa.as_ref().as_slice());
            b.extend_from_slice(fb.as_ref().as_slice());
        }
        let n = (a.len() / 3) as u32;
        if (n as usize) < <eight_point::EightPoint as Estimator<FeatureMatch>>::MIN_SAMPLES {
            return None;
        }
        self.run(false, &a, &b, n).map(|(rt, inl)| (CameraToCamera(isometry(&rt)), inl))
    }
}
/// `Consensus<LambdaTwist, FeatureWorldMatch>` — lambda-twist/src/lib.rs:330-347 estimates, cv-core/src/pose.rs:194-201 scores.
impl Consensus<lambda_twist::LambdaTwist, FeatureWorldMatch> for Arrsac {
    type Inliers = Vec<usize>;
    fn model<I>(&mut self, estimator: &lambda_twist::LambdaTwist, data: I) -> Option<WorldToCamera>
    where I: Iterator<Item = FeatureWorldMatch> + Clone {
        self.model_inliers(estimator, data).map(|(m, _)| m)
    }
    fn model_inliers<I>(&mut self, _estimator: &lambda_twist::LambdaTwist, data: I) -> Option<(WorldToCamera, Vec<usize>)>
    where I: Iterator<Item = FeatureWorldMatch> + Clone {
        let (mut a, mut w) = (Vec::new(), Vec::new());
        for FeatureWorldMatch(bearing, world) in data {
            a.extend_from_slice(bearing.as_ref().as_slice());
            w.extend_from_slice(world.homogeneous().as_slice()); // Projective form: xyz normalised, w = 1 / distance
        }
        let n = (a.len() / 3) as u32;
        if (n as usize) < <lambda_twist::LambdaTwist as Estimator<FeatureWorldMatch>>::MIN_SAMPLES {
            return None;
        }
        self.run(true, &a, &w, n).map(|(rt, inl)| (WorldToCamera(isometry(&rt)), inl))
    }
}
/// `UnitVector3` bearings of a match list, for callers that hold keypoints: `CameraIntrinsics::calibrate` is the reference's
/// own (cv-pinhole/src/lib.rs:108-117) and stays on the host — it is a handful of flops per keypoint.
pub fn bearing(v: [f64; 3]) -> UnitVector3<f64> {
    UnitVector3::new_unchecked(Vector3::new(v[0], v[1], v[2]))
}
