// host_masks.cpp -- host-side (no GPU) variable-density Poisson-disc sampling for `Poisson2DMaskFunc` (reference data/subsample.py:549-633,
// itself adapted from sigpy.mri.samp: Bridson's dart throwing with a per-pixel elliptical exclusion radius).
//
// The reference runs this loop under numba.jit on Numba's private generator, so its masks are not reproducible from any seed even there; this
// routine is the same algorithm on its own splitmix64 / xoshiro256** stream seeded by the caller: same seed, same mask, on every platform.
//   * a calibration rectangle of calib_y x calib_x pixels around the centre is set first;
//   * one random first point; while the active list is not empty: pick a random active point p, try up to max_attempts candidates
//     q = p + v (r_x(p) cos t, r_y(p) sin t) with v^2 uniform in [1, 4) and t uniform in [0, 2 pi); a candidate is accepted when it lies in the
//     grid and no sample (x, y) of its window has ((q_x - x) / r_x(x, y))^2 + ((q_y - y) / r_y(x, y))^2 < 1; an accepted point joins the
//     mask and the active list, a point whose attempts all failed leaves the list.
#include <cmath>
#include <cstdint>
#include <vector>

#include "mrx_common.h"

namespace {
struct Rng {
    uint64_t s[4];
    static uint64_t splitmix(uint64_t& x) {
        uint64_t z = (x += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    explicit Rng(uint64_t seed) {
        for (auto& v : s) v = splitmix(seed);
    }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next() {   // xoshiro256**
        const uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0], s[3] ^= s[1], s[1] ^= s[2], s[0] ^= s[3], s[2] ^= t, s[3] = rotl(s[3], 45);
        return r;
    }
    double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }   // [0, 1)
    int64_t below(int64_t n) { return (int64_t)(uniform() * (double)n); }             // [0, n)
};
}  // namespace

// mask: ny x nx bytes (row-major), overwritten with 0 / 1; radius_x / radius_y: ny x nx floats (>= 1).  Returns the number of samples or a
// negative MRX_E* code.
extern "C" int64_t mrx_poisson_disc_mask(int nx, int ny, int max_attempts, const float* radius_x, const float* radius_y, double calib_x, double calib_y,
                                         uint64_t seed, unsigned char* mask) {
    if (!(nx >= 1 && ny >= 1 && max_attempts >= 1 && radius_x && radius_y && mask)) {
        mrx_set_error("mrx_poisson_disc_mask: bad argument");
        return MRX_EINVAL;
    }
    const int64_t n = (int64_t)nx * ny;
    for (int64_t i = 0; i < n; ++i) mask[i] = 0;
    // the calibration block (subsample.py:575-578: int() truncation of centre -/+ half the block)
    {
        const int y0 = (int)(ny / 2.0 - calib_y / 2.0), y1 = (int)(ny / 2.0 + calib_y / 2.0);
        const int x0 = (int)(nx / 2.0 - calib_x / 2.0), x1 = (int)(nx / 2.0 + calib_x / 2.0);
        for (int y = y0 < 0 ? 0 : y0; y < y1 && y < ny; ++y)
            for (int x = x0 < 0 ? 0 : x0; x < x1 && x < nx; ++x) mask[(int64_t)y * nx + x] = 1;
    }
    Rng rng(seed);
    std::vector<int> px((size_t)n + 1), py((size_t)n + 1);
    px[0] = (int)rng.below(nx), py[0] = (int)rng.below(ny);
    int64_t active = 1;
    const double two_pi = 6.283185307179586476925286766559;
    while (active > 0) {
        const int64_t i = rng.below(active);
        const int cx = px[(size_t)i], cy = py[(size_t)i];
        const double rx = radius_x[(int64_t)cy * nx + cx], ry = radius_y[(int64_t)cy * nx + cx];
        bool placed = false;
        double qx = 0, qy = 0;
        for (int k = 0; k < max_attempts && !placed; ++k) {
            const double v = std::sqrt(rng.uniform() * 3.0 + 1.0), t = two_pi * rng.uniform();
            qx = cx + v * rx * std::cos(t), qy = cy + v * ry * std::sin(t);
            if (!(qx >= 0 && qx < nx && qy >= 0 && qy < ny)) continue;
            const int x0 = (int)(qx - rx) < 0 ? 0 : (int)(qx - rx), x1 = (int)(qx + rx + 1) > nx ? nx : (int)(qx + rx + 1);
            const int y0 = (int)(qy - ry) < 0 ? 0 : (int)(qy - ry), y1 = (int)(qy + ry + 1) > ny ? ny : (int)(qy + ry + 1);
            placed = true;
            for (int y = y0; y < y1 && placed; ++y)
                for (int x = x0; x < x1; ++x) {
                    if (!mask[(int64_t)y * nx + x]) continue;
                    const double dx = (qx - x) / radius_x[(int64_t)y * nx + x], dy = (qy - y) / radius_y[(int64_t)y * nx + x];
                    if (dx * dx + dy * dy < 1.0) {
                        placed = false;
                        break;
                    }
                }
        }
        if (placed && active <= n) {
            px[(size_t)active] = (int)qx, py[(size_t)active] = (int)qy;
            mask[(int64_t)(int)qy * nx + (int)qx] = 1;
            ++active;
            if (active > n) break;   // cannot happen for radii >= 1 (every accepted point occupies a cell of its own); a guard against bad radii
        } else {
            px[(size_t)i] = px[(size_t)(active - 1)], py[(size_t)i] = py[(size_t)(active - 1)];
            --active;
        }
    }
    int64_t count = 0;
    for (int64_t i = 0; i < n; ++i) count += mask[i];
    return count;
}
