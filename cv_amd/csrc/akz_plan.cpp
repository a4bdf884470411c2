// akz_plan.cpp — host-only scalar math of the AKAZE pyramid schedule.
//
// Follows akaze/src/evolution.rs:46-126 (EvolutionStep::new, Akaze::allocate_evolutions),
// akaze/src/fed_tau.rs:26-93 (FED cycle step sizes with kappa-cycle reordering),
// akaze/src/image.rs:349-389 (gaussian, gaussian_kernel, kernel radius) and
// akaze/src/derivatives.rs:57-79 (multiscale Scharr weights).  All of it is f64/f32 scalar
// arithmetic evaluated on the host with the host libm, exactly as the reference does.
#include <math.h>

#include "akz_common.h"

namespace {

bool is_prime(uint64_t n)
{
    if (n < 2) return false;
    for (uint64_t d = 2; d * d <= n; ++d)
        if (n % d == 0) return false;
    return true;
}

// `x as usize` on f64: truncation toward zero, saturating at 0 (NaN -> 0).
size_t to_usize(double v) { return v > 0.0 ? (v >= 1.8446744073709552e19 ? SIZE_MAX : (size_t)v) : 0; }

// fed_tau_by_process_time(T, 1, tau_max, true) — fed_tau.rs:26-93.
std::vector<double> fed_tau_cycle(double T, double tau_max)
{
    const double kPi = 3.14159265358979323846;
    // fed_tau_by_cycle_time, :44-46
    size_t n = to_usize(ceil(sqrt(3.0 * T / tau_max + 0.25) - 0.5 - 1.0e-8) + 0.5);
    double scale = 3.0 * T / (tau_max * (double)(n * (n + 1)));
    // fed_tau_internal, :60-68
    std::vector<double> tau(n);
    for (size_t k = 0; k < n; ++k) {
        double c = 1.0 / (4.0 * (double)n + 2.0);
        double d = scale * tau_max / 2.0;
        double hcos = cos(kPi * (2.0 * (double)k + 1.0) * c);
        tau[k] = d / (hcos * hcos);
    }
    if (n == 0) return tau;
    // kappa-cycle reordering, :69-89
    size_t kappa = n / 2;
    size_t prime = n + 1;
    while (!is_prime(prime)) ++prime;
    std::vector<double> out(n);
    size_t k = 0;
    for (size_t l = 0; l < n; ++l) {
        size_t index = ((k + 1) * kappa) % prime - 1;  // usize wrap-around as in a release build
        while (index >= n) {
            ++k;
            index = ((k + 1) * kappa) % prime - 1;
        }
        ++k;
        out[l] = tau[index];
    }
    return out;
}

}  // namespace

void akz_build_plan(const akz_config& cfg, int w, int h, AkzPlan* plan)
{
    plan->w = w;
    plan->h = h;
    plan->levels.clear();
    plan->n_octaves = 0;
    int lw = w, lh = h;
    int prev_octave = -1;
    for (uint32_t octave = 0; octave < cfg.max_octave_evolution; ++octave) {
        double rfactor = ldexp(1.0, -(int)octave);  // 2.0f64.powi(-octave)
        uint32_t level_h = (uint32_t)((double)h * rfactor);
        uint32_t level_w = (uint32_t)((double)w * rfactor);
        uint32_t smallest = level_w < level_h ? level_w : level_h;
        if (octave > 0) {
            lw /= 2;  // what GrayFloatImage::half_size produces (image.rs:155-156)
            lh /= 2;
        }
        if (smallest < 40) continue;  // evolution.rs:89-90 (filter_map skips; later octaves are smaller)
        uint32_t sublevels = smallest < 80 ? 1u : cfg.num_sublevels;
        for (uint32_t s = 0; s < sublevels; ++s) {
            AkzLevel L;
            L.w = lw;
            L.h = lh;
            L.octave = octave;
            L.sublevel = s;
            L.esigma = cfg.base_scale_offset *
                       pow(2.0, (double)s / (double)cfg.num_sublevels + (double)octave);
            L.etime = 0.5 * (L.esigma * L.esigma);
            double ratio = ldexp(1.0, (int)octave);
            double sig = round(L.esigma * cfg.derivative_factor / ratio);
            L.deriv_sigma = (uint32_t)sig;
            L.sigma_quat = (float)(sig * sig * sig * sig);
            L.kp_size = (float)(L.esigma * cfg.derivative_factor);
            L.new_octave = !plan->levels.empty() && (int)octave > prev_octave;
            prev_octave = (int)octave;
            plan->levels.push_back(L);
        }
    }
    for (size_t i = 1; i < plan->levels.size(); ++i) {
        double ttime = plan->levels[i].etime - plan->levels[i - 1].etime;
        plan->levels[i].tau = fed_tau_cycle(ttime / 1.0, 0.25);
    }
    // The pixels that are interior (scale_space_extrema.rs:50) AND pass the border test (:96-104) as inclusive
    // integer ranges: the test is monotone in x and in y, so the reference's f32 expressions are evaluated once per
    // column / row here instead of per candidate on the device (the streaming determinant kernel compares integers).
    for (auto& L : plan->levels) {
        const float ratio = ldexpf(1.0f, (int)L.octave);
        const float sigma_size = roundf(L.kp_size / ratio);          // scale_space_extrema.rs:69-70
        const float smax = 10.0f * sqrtf(2.0f);                      // :16
        L.cand_border = smax * sigma_size;
        auto ok = [&](int p, int extent) {
            const float f = (float)p;
            volatile float lo = f - L.cand_border, hi = f + L.cand_border;   // one f32 rounding each, as on the device
            return !(roundf(lo) - 1.0f < 0.0f) && !(roundf(hi) + 1.0f >= (float)extent);
        };
        L.cand_x_lo = L.cand_y_lo = 1 << 30;
        L.cand_x_hi = L.cand_y_hi = -1;
        for (int x = 1; x <= L.w - 2; ++x)
            if (ok(x, L.w)) {
                if (x < L.cand_x_lo) L.cand_x_lo = x;
                L.cand_x_hi = x;
            }
        for (int y = 1; y <= L.h - 2; ++y)
            if (ok(y, L.h)) {
                if (y < L.cand_y_lo) L.cand_y_lo = y;
                L.cand_y_hi = y;
            }
    }
    plan->sum_pixels = 0;
    for (auto& L : plan->levels) {
        plan->sum_pixels += L.pixels();
        if ((int)L.octave + 1 > plan->n_octaves) plan->n_octaves = (int)L.octave + 1;
    }
}

int akz_gaussian_radius(float r) { return (int)to_usize((double)ceilf(2.0f * r)); }

void akz_host_gaussian_kernel(float r, int ksize, float* out)
{
    const float kPiF = 3.14159274101257324219f;
    int half = ksize / 2;
    float sum = 0.0f;
    for (int i = -half; i <= half; ++i) {
        float x = (float)i;
        float norm = 1.0f / (sqrtf(2.0f * kPiF) * r);
        float val = norm * expf(-(x * x) / (2.0f * (r * r)));
        out[i + half] = val;
        sum += val;
    }
    for (int i = 0; i < ksize; ++i) out[i] /= sum;
}

ScharrW akz_scharr_weights(uint32_t sigma)
{
    ScharrW s;
    double w = 10.0 / 3.0;
    s.norm = (float)(1.0 / (2.0 * (double)sigma * (w + 2.0)));
    s.middle = s.norm * (float)w;
    s.sigma = (int)sigma;
    return s;
}
