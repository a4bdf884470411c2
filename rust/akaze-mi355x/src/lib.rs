//! `akaze`-compatible front-end over the MI355X library (C ABI declared in `include/akz.h`).
//!
//! NOT BUILT in this repository (no Rust toolchain in the build image); kept as the reference-side
//! binding of INTEGRATION.md.  The public surface is the one rust-cv callers use
//! (`akaze/src/lib.rs:69-185,295-366` of rust-cv/cv): `Akaze` with its 11 public fields,
//! `Akaze::{new, sparse, dense, extract, extract_from_gray_float_image, extract_path}`, `KeyPoint`, the `image`
//! module (`GrayFloatImage`, the separable filters, `gaussian_kernel`), a `space::Knn` implementor, the two-view
//! consensus and the place-recognition hasher.
use bitarray::BitArray;
use cv_core::{nalgebra::Point2, ImagePoint};
use ::image::{DynamicImage, ImageResult};
use std::{os::raw::c_void, path::Path, ptr};

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

extern "C" {
    fn akz_create(cfg: *const AkzConfig, device: i32, max_w: i32, max_h: i32, max_batch: i32, max_kp: u32,
                  out: *mut *mut c_void) -> i32;
    fn akz_destroy(ctx: *mut c_void) -> i32;
    fn akz_extract_gray_u16(ctx: *mut c_void, img: *const u16, w: i32, h: i32, stride: i32, kps: *mut AkzKeypoint,
                            descs: *mut [u8; 64], cap: u32, n_out: *mut u32) -> i32;
    fn akz_extract_gray_u8(ctx: *mut c_void, img: *const u8, w: i32, h: i32, stride: i32, kps: *mut AkzKeypoint,
                           descs: *mut [u8; 64], cap: u32, n_out: *mut u32) -> i32;
    fn akz_extract_gray_f32(ctx: *mut c_void, img: *const f32, w: i32, h: i32, stride: i32, kps: *mut AkzKeypoint,
                            descs: *mut [u8; 64], cap: u32, n_out: *mut u32) -> i32;
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
    fn hm_create(device: i32, max_q: u32, max_t: u32, out: *mut *mut c_void) -> i32;
    fn hm_destroy(ctx: *mut c_void) -> i32;
    fn hm_knn2(ctx: *mut c_void, q: *const [u8; 64], nq: u32, t: *const [u8; 64], nt: u32, out: *mut AkzNeighbor) -> i32;
    fn hm_knn(ctx: *mut c_void, q: *const [u8; 64], nq: u32, t: *const [u8; 64], nt: u32, k: u32,
              out: *mut AkzNeighbor) -> i32;
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

const MAX_KP: u32 = 16384;

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

    fn run<F: FnOnce(*mut c_void, *mut AkzKeypoint, *mut [u8; 64], *mut u32) -> i32>(
        &self, w: u32, h: u32, f: F,
    ) -> (Vec<KeyPoint>, Vec<BitArray<64>>) {
        // `Akaze` is Copy and stateless in the reference, so the device context is created per call here;
        // a production shim would cache one context per (config, size) in a thread-local.
        let cfg = self.config();
        let mut ctx: *mut c_void = ptr::null_mut();
        let st = unsafe { akz_create(&cfg, 0, w as i32, h as i32, 1, MAX_KP, &mut ctx) };
        assert_eq!(st, 0, "akz_create failed with status {st} (there is no CPU fallback)");
        let mut kps = vec![AkzKeypoint::default(); MAX_KP as usize];
        let mut descs = vec![[0u8; 64]; MAX_KP as usize];
        let mut n = 0u32;
        let st = f(ctx, kps.as_mut_ptr(), descs.as_mut_ptr(), &mut n);
        unsafe { akz_destroy(ctx) };
        assert_eq!(st, 0, "akz_extract failed with status {st}");
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
        (keypoints, descriptors)
    }

    /// `Akaze::extract` (akaze/src/lib.rs:295).
    pub fn extract(&self, image: &DynamicImage) -> (Vec<KeyPoint>, Vec<BitArray<64>>) {
        match image.grayscale() {
            DynamicImage::ImageLuma8(g) => {
                let (w, h) = (g.width(), g.height());
                self.run(w, h, |ctx, k, d, n| unsafe {
                    akz_extract_gray_u8(ctx, g.as_raw().as_ptr(), w as i32, h as i32, w as i32, k, d, MAX_KP, n)
                })
            }
            DynamicImage::ImageLuma16(g) => {
                // image.rs:57-66: f32::from(v) / 65535f32, done on the device
                let (w, h) = (g.width(), g.height());
                self.run(w, h, |ctx, k, d, n| unsafe {
                    akz_extract_gray_u16(ctx, g.as_raw().as_ptr(), w as i32, h as i32, w as i32, k, d, MAX_KP, n)
                })
            }
            other => {
                // remaining arms of GrayFloatImage::from_dynamic (image.rs:67-106): convert on the host
                let f = other.to_luma32f();
                let (w, h) = (f.width(), f.height());
                self.run(w, h, |ctx, k, d, n| unsafe {
                    akz_extract_gray_f32(ctx, f.as_raw().as_ptr(), w as i32, h as i32, w as i32, k, d, MAX_KP, n)
                })
            }
        }
    }

    /// `Akaze::extract_from_gray_float_image` (akaze/src/lib.rs:309).
    pub fn extract_from_gray_float_image(&self, img: &crate::image::GrayFloatImage) -> (Vec<KeyPoint>, Vec<BitArray<64>>) {
        let (w, h) = (img.width() as u32, img.height() as u32);
        self.run(w, h, |ctx, k, d, n| unsafe {
            akz_extract_gray_f32(ctx, img.0.as_raw().as_ptr(), w as i32, h as i32, w as i32, k, d, MAX_KP, n)
        })
    }

    /// `Akaze::extract_path` (akaze/src/lib.rs:361).
    pub fn extract_path(&self, path: impl AsRef<Path>) -> ImageResult<(Vec<KeyPoint>, Vec<BitArray<64>>)> {
        Ok(self.extract(&::image::open(path)?))
    }
}

/// A `space::Knn` implementor with `LinearKnn { metric: Hamming, iter }` semantics for `BitArray<64>`.
pub struct Mi355xLinearKnn<'a> {
    pub targets: &'a [BitArray<64>],
}
impl<'a> space::Knn for Mi355xLinearKnn<'a> {
    type Ix = usize;
    type Metric = bitarray::Hamming;
    type Point = BitArray<64>;
    type KnnIter = Vec<space::Neighbor<u32, usize>>;
    fn knn(&self, query: &BitArray<64>, num: usize) -> Self::KnnIter {
        // cv-sfm asks for 2 when matching frame pairs (lib.rs:3103) and 3 when registering a frame (lib.rs:1474)
        assert!((1..=3).contains(&num), "the MI355X matcher implements knn(query, k) for k <= 3");
        let mut ctx: *mut c_void = ptr::null_mut();
        let n = self.targets.len() as u32;
        assert_eq!(unsafe { hm_create(0, 1, n.max(2), &mut ctx) }, 0);
        let mut out = [AkzNeighbor { index: 0, distance: 0 }; 3];
        let st = unsafe {
            hm_knn(ctx, query.bytes() as *const [u8; 64], 1, self.targets.as_ptr() as *const [u8; 64], n, num as u32,
                   out.as_mut_ptr())
        };
        unsafe { hm_destroy(ctx) };
        assert_eq!(st, 0);
        // LinearKnn returns min(num, len) neighbours
        out.iter().take(num.min(self.targets.len()))
            .map(|o| space::Neighbor { index: o.index as usize, distance: o.distance }).collect()
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
        let mut ctx: *mut c_void = ptr::null_mut();
        let cap = (feats.len() as u32).max(self.codewords.len() as u32);
        assert_eq!(unsafe { hm_create(0, cap, cap, &mut ctx) }, 0);
        let mut hash = [0u8; 512];
        let st = unsafe {
            hm_hash_bag(ctx, feats.as_ptr() as *const [u8; 64], feats.len() as u32,
                        self.codewords.as_ptr() as *const [u8; 64], self.codewords.len() as u32, hash.as_mut_ptr(),
                        ptr::null_mut())
        };
        unsafe { hm_destroy(ctx) };
        assert_eq!(st, 0);
        BitArray::new(hash)
    }
}

/// `lsh_to_frame.knn_values(&lsh, num)` (cv-sfm/src/lib.rs:622-624) as an exact search over the stored hashes:
/// (index into `hashes`, distance), ascending (distance, index).
pub fn nearest_hashes(query: &BitArray<512>, hashes: &[BitArray<512>], num: usize) -> Vec<(usize, u32)> {
    let mut ctx: *mut c_void = ptr::null_mut();
    assert_eq!(unsafe { hm_create(0, 2, 2, &mut ctx) }, 0);
    let mut out = vec![AkzNeighbor { index: 0, distance: 0 }; num.max(1)];
    let mut n: u32 = 0;
    let st = unsafe {
        hm_hash_knn(ctx, query.bytes().as_ptr(), hashes.as_ptr() as *const u8, hashes.len() as u32, 512, num as u32,
                    out.as_mut_ptr(), &mut n)
    };
    unsafe { hm_destroy(ctx) };
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

/// Two-view consensus on the MI355X: what `Consensus::model_inliers(&EightPoint::new(), matches)` computes
/// (akaze/tests/estimate_pose.rs:63-67, tutorial ch5 main.rs:70-72, cv-sfm/src/lib.rs:1394-1406), with the
/// minimal samples drawn by the caller's RNG (`arrsac` draws them inside; the crate is not part of this
/// repository).  `bearings_a[i]` / `bearings_b[i]` are the unit bearings of match i; `samples` holds 8 match
/// indices per hypothesis.  Returns the winning pose as a row-major 3x4 `[R | t]` and the inlier indices.
pub struct Mi355xEssentialConsensus {
    pub inlier_threshold: f64,
}
impl Mi355xEssentialConsensus {
    pub fn model_inliers(&mut self, bearings_a: &[[f64; 3]], bearings_b: &[[f64; 3]], samples: &[[u32; 8]])
        -> Option<([f64; 12], Vec<usize>)> {
        assert_eq!(bearings_a.len(), bearings_b.len());
        let n = bearings_a.len() as u32;
        let mut ctx: *mut c_void = ptr::null_mut();
        assert_eq!(unsafe { rs_create(0, n.max(8), samples.len().max(1) as u32, &mut ctx) }, 0);
        let mut pose = [0f64; 12];
        let (mut best, mut n_inl) = (0u32, 0u32);
        let mut inl = vec![0u32; n as usize];
        let st = unsafe {
            rs_essential_batch(ctx, bearings_a.as_ptr() as *const f64, bearings_b.as_ptr() as *const f64, n,
                               samples.as_ptr() as *const u32, samples.len() as u32, self.inlier_threshold,
                               pose.as_mut_ptr(), &mut best, inl.as_mut_ptr(), n, &mut n_inl)
        };
        unsafe { rs_destroy(ctx) };
        assert_eq!(st, 0);
        if best == u32::MAX {
            return None;
        }
        Some((pose, inl[..n_inl as usize].iter().map(|&i| i as usize).collect()))
    }
}

