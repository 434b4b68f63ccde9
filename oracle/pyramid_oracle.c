/*
 * oracle/pyramid_oracle.c -- CPU restatement of shinestacker's Laplacian-pyramid
 * focus-stacking arithmetic.  TEST INFRASTRUCTURE ONLY: nothing under
 * shinestacker_amd/ may link, import or call this file.  It is the checker for
 * the HIP path (tests/, __graft_entry__.smoke, bench.py's cpu_baseline leg).
 *
 * Reference followed (paths relative to /root/reference/src/shinestacker):
 *   algorithms/pyramid.py:11-22   generating kernel (orc_make_kernel in oracle.py)
 *   algorithms/pyramid.py:24-25   convolve  = cv2.filter2D(BORDER_REFLECT101)
 *   algorithms/pyramid.py:27-32   reduce_layer
 *   algorithms/pyramid.py:34-46   expand_layer (zero-stuff, 4*convolve)
 *   algorithms/pyramid.py:48-55   fuse_laplacian (gray, square, convolve, argmax)
 *   algorithms/pyramid.py:57-64   collapse
 *   algorithms/pyramid.py:66-111  base-level entropy / deviation fusion
 *   algorithms/pyramid.py:125-139 process_single_image
 *
 * PARITY STATUS: "parity unpinned" for the three cv2 primitives.  OpenCV
 * (opencv_python, unpinned, pyproject.toml:26) is neither vendored in the
 * reference nor installed here, and the reference tests hold no numeric
 * vectors for this path.  The primitives below restate OpenCV's published
 * direct-filter algorithm [from memory]:
 *   filter2D, CV_32F, 5x5 kernel: kernel converted to float32, non-zero taps
 *     visited in row-major order, s = delta(0); s = fma(k, x, s) per tap
 *     (AVX2/FMA3 dispatch) or s += k*x (SSE baseline) -> `use_fma` flag.
 *   cvtColor BGR2GRAY float: fma(R,.299f, fma(G,.587f, B*.114f)).
 *   BORDER_REFLECT101: gfedcb|abcdefgh|gfedcba.
 * The NumPy-side control flow of the reference IS pinned: oracle/gen_golden.py
 * imports the reference's own pyramid.py with these primitives injected as a
 * cv2 shim and freezes its outputs under tests/golden/.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#if defined(_OPENMP)
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

static inline int r101(int i, int n) {
    /* BORDER_REFLECT101 for |overshoot| < n */
    if (n == 1) return 0;
    while (i < 0 || i >= n) {
        if (i < 0) i = -i;
        if (i >= n) i = 2 * n - 2 - i;
    }
    return i;
}

static inline float mac32(float k, float x, float s, int use_fma) {
    if (use_fma) return __builtin_fmaf(k, x, s);
    volatile float p = k * x; /* volatile: forbid contraction whatever the flags */
    return s + p;
}

static inline double mac64(double k, double x, double s, int use_fma) {
    if (use_fma) return __builtin_fma(k, x, s);
    volatile double p = k * x;
    return s + p;
}

ORC_API void orc_set_threads(int n) {
#if defined(_OPENMP)
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

ORC_API int orc_num_threads(void) {
#if defined(_OPENMP)
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ---- cv2.filter2D(img, -1, K5x5, borderType=REFLECT101), single channel ---- */
ORC_API void orc_filter2d_f32(const float* src, int h, int w, const float* k25,
                              float* dst, int use_fma) {
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y) {
        int ry[5];
        for (int t = 0; t < 5; ++t) ry[t] = r101(y + t - 2, h);
        for (int x = 0; x < w; ++x) {
            float s = 0.0f;
            for (int ty = 0; ty < 5; ++ty) {
                const float* row = src + (size_t)ry[ty] * w;
                for (int tx = 0; tx < 5; ++tx) {
                    float kk = k25[ty * 5 + tx];
                    if (kk == 0.0f) continue; /* OpenCV keeps only non-zero taps */
                    s = mac32(kk, row[r101(x + tx - 2, w)], s, use_fma);
                }
            }
            dst[(size_t)y * w + x] = s;
        }
    }
}

ORC_API void orc_filter2d_f64(const double* src, int h, int w, const double* k25,
                              double* dst, int use_fma) {
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y) {
        int ry[5];
        for (int t = 0; t < 5; ++t) ry[t] = r101(y + t - 2, h);
        for (int x = 0; x < w; ++x) {
            double s = 0.0;
            for (int ty = 0; ty < 5; ++ty) {
                const double* row = src + (size_t)ry[ty] * w;
                for (int tx = 0; tx < 5; ++tx) {
                    double kk = k25[ty * 5 + tx];
                    if (kk == 0.0) continue;
                    s = mac64(kk, row[r101(x + tx - 2, w)], s, use_fma);
                }
            }
            dst[(size_t)y * w + x] = s;
        }
    }
}

/* ---- cv2.cvtColor(BGR2GRAY) on float32 ---- */
static inline float gray_of(float b, float g, float r, int use_fma) {
    const float cb = 0.114f, cg = 0.587f, cr = 0.299f;
    if (use_fma) return __builtin_fmaf(r, cr, __builtin_fmaf(g, cg, b * cb));
    volatile float pb = b * cb, pg = g * cg, pr = r * cr;
    volatile float s = pb + pg;
    return s + pr;
}

ORC_API void orc_bgr2gray_f32(const float* bgr, size_t npix, float* gray, int use_fma) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < npix; ++i)
        gray[i] = gray_of(bgr[3 * i], bgr[3 * i + 1], bgr[3 * i + 2], use_fma);
}

/* =====================================================================
 * Streaming (decimated / polyphase) restatement.  Same arithmetic as the
 * reference-shaped flow -- every skipped tap is an exact +0 product of the
 * zero-stuffed grid -- but O(1) memory in the number of frames.
 * ===================================================================== */

/* reduce_layer, pyramid.py:27-32: conv at full res then [::2, ::2]  ==
 * conv evaluated at even coordinates only.  Interleaved C channels. */
ORC_API void orc_reduce_f32(const float* g, int h, int w, int C, const float* k25,
                            float* out, int use_fma) {
    int ho = (h + 1) / 2, wo = (w + 1) / 2;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < ho; ++i) {
        int ry[5];
        for (int t = 0; t < 5; ++t) ry[t] = r101(2 * i + t - 2, h);
        for (int j = 0; j < wo; ++j) {
            int rx[5];
            if (2 * j - 2 >= 0 && 2 * j + 2 < w) {
                for (int t = 0; t < 5; ++t) rx[t] = 2 * j + t - 2;
            } else {
                for (int t = 0; t < 5; ++t) rx[t] = r101(2 * j + t - 2, w);
            }
            for (int c = 0; c < C; ++c) {
                float s = 0.0f;
                for (int ty = 0; ty < 5; ++ty) {
                    const float* row = g + (size_t)ry[ty] * w * C + c;
                    for (int tx = 0; tx < 5; ++tx)
                        s = mac32(k25[ty * 5 + tx], row[(size_t)rx[tx] * C], s, use_fma);
                }
                out[((size_t)i * wo + j) * C + c] = s;
            }
        }
    }
}

/* value of expand_layer(src)[y, x, c] (pyramid.py:34-46): src is hs x ws x C,
 * the zero-stuffed grid is 2hs x 2ws, REFLECT101 acts on THAT grid. */
static inline float expand_at(const float* src, int hs, int ws, int C, const float* k25,
                              int y, int x, int c, int use_fma) {
    float s = 0.0f;
    int H2 = 2 * hs, W2 = 2 * ws;
    if (y >= 2 && y + 2 < H2 && x >= 2 && x + 2 < W2) {
        /* interior: no reflection; the non-stuffed taps are those with (y+ty) and (x+tx) even */
        const int ty0 = y & 1, tx0 = x & 1; /* first tap index with even target coordinate */
        for (int ty = ty0; ty < 5; ty += 2) {
            const float* row = src + ((size_t)((y + ty - 2) >> 1) * ws) * C + c;
            for (int tx = tx0; tx < 5; tx += 2)
                s = mac32(k25[ty * 5 + tx], row[(size_t)((x + tx - 2) >> 1) * C], s, use_fma);
        }
        return 4.0f * s;
    }
    for (int ty = 0; ty < 5; ++ty) {
        int yy = r101(y + ty - 2, H2);
        if (yy & 1) continue; /* stuffed zero row */
        for (int tx = 0; tx < 5; ++tx) {
            int xx = r101(x + tx - 2, W2);
            if (xx & 1) continue;
            s = mac32(k25[ty * 5 + tx], src[((size_t)(yy >> 1) * ws + (xx >> 1)) * C + c], s,
                      use_fma);
        }
    }
    return 4.0f * s;
}

ORC_API void orc_expand_f32(const float* src, int hs, int ws, int C, const float* k25,
                            int h, int w, float* out, int use_fma) {
    /* out is h x w x C with h <= 2hs, w <= 2ws (the caller's crop) */
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x)
            for (int c = 0; c < C; ++c)
                out[((size_t)y * w + x) * C + c] = expand_at(src, hs, ws, C, k25, y, x, c, use_fma);
}

/* Laplacian level + focus energy + running first-max selection for one frame.
 * g: level l (h x w x 3), gn: level l+1.  best_e/best_lap/best_idx are the
 * running state (pyramid.py:48-55 restated as a stream: strict '>' keeps the
 * first maximum when frames arrive in ascending index order).
 * scratch: caller-provided h*w*4 floats (lap, then q). first!=0 initialises. */
ORC_API void orc_level_select_f32(const float* g, int h, int w, const float* gn, int hs, int ws,
                                  const float* k25, int frame_idx, int first, float* best_e,
                                  float* best_lap, int32_t* best_idx, float* scratch,
                                  int use_fma) {
    float* lap = scratch;
    float* q = scratch + (size_t)h * w * 3;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            float v[3];
            for (int c = 0; c < 3; ++c) {
                float e = expand_at(gn, hs, ws, 3, k25, y, x, c, use_fma);
                v[c] = g[((size_t)y * w + x) * 3 + c] - e;
                lap[((size_t)y * w + x) * 3 + c] = v[c];
            }
            float gr = gray_of(v[0], v[1], v[2], use_fma);
            q[(size_t)y * w + x] = gr * gr;
        }
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y) {
        int ry[5];
        for (int t = 0; t < 5; ++t) ry[t] = r101(y + t - 2, h);
        for (int x = 0; x < w; ++x) {
            float s = 0.0f;
            if (x >= 2 && x + 2 < w) {
                for (int ty = 0; ty < 5; ++ty) {
                    const float* row = q + (size_t)ry[ty] * w + x - 2;
                    for (int tx = 0; tx < 5; ++tx) s = mac32(k25[ty * 5 + tx], row[tx], s, use_fma);
                }
            } else {
                for (int ty = 0; ty < 5; ++ty)
                    for (int tx = 0; tx < 5; ++tx)
                        s = mac32(k25[ty * 5 + tx], q[(size_t)ry[ty] * w + r101(x + tx - 2, w)], s,
                                  use_fma);
            }
            size_t p = (size_t)y * w + x;
            if (first || s > best_e[p]) {
                best_e[p] = s;
                best_idx[p] = frame_idx;
                for (int c = 0; c < 3; ++c) {
                    float lv = lap[p * 3 + c];
                    best_lap[p * 3 + c] = (lv == 0.0f) ? 0.0f : lv; /* -0 -> +0, np.where sum */
                }
            }
        }
    }
}

/* collapse step, pyramid.py:59-63: img_l = expand(img_{l+1})[:h,:w] + lap_l */
ORC_API void orc_collapse_level_f32(const float* up, int hs, int ws, const float* k25,
                                    const float* lap, int h, int w, float* out, int use_fma) {
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x)
            for (int c = 0; c < 3; ++c) {
                size_t p = ((size_t)y * w + x) * 3 + c;
                out[p] = expand_at(up, hs, ws, 3, k25, y, x, c, use_fma) + lap[p];
            }
}

/* pyramid.py:64 + :179  clip(abs(img), 0, max).astype(dtype)  (truncation) */
ORC_API void orc_finalize_u8(const float* img, size_t n, uint8_t* out) {
    for (size_t i = 0; i < n; ++i) {
        float v = fabsf(img[i]);
        if (v > 255.0f) v = 255.0f;
        out[i] = (uint8_t)v;
    }
}
ORC_API void orc_finalize_u16(const float* img, size_t n, uint16_t* out) {
    for (size_t i = 0; i < n; ++i) {
        float v = fabsf(img[i]);
        if (v > 65535.0f) v = 65535.0f;
        out[i] = (uint16_t)v;
    }
}

/* ---- NumPy float32 add.reduce order on a contiguous array --------------------
 * np.add.reduce hands all n elements to the pairwise loop (8 accumulators,
 * blocks of 128, recursive halving above).  Checked bit-for-bit against
 * numpy 2.2.6 in tests/test_oracle.py (1-D and coalesced 2-D inputs). */
static float np_pairwise_f32(const float* a, int n) {
    if (n < 8) {
        float res = -0.0f;
        for (int i = 0; i < n; ++i) res += a[i];
        return res;
    } else if (n <= 128) {
        float r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        int i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    } else {
        int n2 = n / 2;
        n2 -= n2 % 8;
        return np_pairwise_f32(a, n2) + np_pairwise_f32(a + n2, n - n2);
    }
}
ORC_API float orc_np_sum_f32(const float* a, int n) {
    if (n == 0) return 0.0f;
    return np_pairwise_f32(a, n);
}

/* ---- base level features, pyramid.py:66-93, 95-102 -------------------------
 * base: hb x wb x 3 float32 BGR (G_L of one frame). pad = (kernel_size-1)//2.
 * nlevels = 256 or 65536.  ent/dev: hb x wb float32.
 * log: float32(log(double(p))) -- a correctly rounded float32 log.  NumPy's own
 * SIMD float32 log differs from this by <=1 ULP on ~20% of inputs and is
 * itself CPU-dispatch dependent, so this term of the reference is not
 * bit-reproducible even between two hosts running the reference. */
ORC_API void orc_base_features_f32(const float* base, int hb, int wb, int nlevels, int pad,
                                   float* ent, float* dev, int use_fma) {
    size_t np_ = (size_t)hb * wb;
    int32_t* lev = (int32_t*)malloc(np_ * sizeof(int32_t));
    uint32_t* cnt = (uint32_t*)calloc((size_t)nlevels, sizeof(uint32_t));
    float* logp = (float*)malloc((size_t)nlevels * sizeof(float));
    for (size_t i = 0; i < np_; ++i) {
        float gr = gray_of(base[3 * i], base[3 * i + 1], base[3 * i + 2], use_fma);
        int32_t l = (int32_t)gr; /* .astype(uint8/uint16): truncation */
        if (l < 0) l = 0;
        if (l >= nlevels) l = nlevels - 1;
        lev[i] = l;
        cnt[l]++;
    }
    for (int l = 0; l < nlevels; ++l) {
        if (cnt[l]) {
            /* counts.astype(float32) / counts.sum()  -> float64 division (NumPy 2
             * promotion with an int64 scalar), stored into a float32 table */
            float p = (float)((double)(float)cnt[l] / (double)np_);
            logp[l] = (float)log((double)p);
        } else
            logp[l] = 0.0f;
    }
    int win = 2 * pad + 1, n = win * win;
    float* buf = (float*)malloc((size_t)n * sizeof(float));
    for (int y = 0; y < hb; ++y)
        for (int x = 0; x < wb; ++x) {
            /* entropy: -(levels * log(p[levels])).sum() */
            double isum = 0.0;
            int t = 0;
            for (int dy = -pad; dy <= pad; ++dy)
                for (int dx = -pad; dx <= pad; ++dx) {
                    int l = lev[(size_t)r101(y + dy, hb) * wb + r101(x + dx, wb)];
                    buf[t++] = (float)l * logp[l];
                    isum += (double)l;
                }
            ent[(size_t)y * wb + x] = -1.0f * orc_np_sum_f32(buf, n);
            /* deviation: square(area - float32(mean)).sum() / area.size */
            float mean = (float)(isum / (double)n);
            t = 0;
            for (int dy = -pad; dy <= pad; ++dy)
                for (int dx = -pad; dx <= pad; ++dx) {
                    int l = lev[(size_t)r101(y + dy, hb) * wb + r101(x + dx, wb)];
                    float d = (float)l - mean;
                    buf[t++] = d * d;
                }
            dev[(size_t)y * wb + x] = orc_np_sum_f32(buf, n) / (float)n;
        }
    free(buf);
    free(logp);
    free(cnt);
    free(lev);
}

/* streaming first-max update for the base features of frame `frame_idx` and
 * the final (img[best_e] + img[best_d]) / 2  (pyramid.py:103-111) */
ORC_API void orc_base_select_f32(const float* ent, const float* dev, size_t npix, int frame_idx,
                                 int first, float* best_ent, float* best_dev, int32_t* idx_e,
                                 int32_t* idx_d) {
    for (size_t i = 0; i < npix; ++i) {
        if (first || ent[i] > best_ent[i]) { best_ent[i] = ent[i]; idx_e[i] = frame_idx; }
        if (first || dev[i] > best_dev[i]) { best_dev[i] = dev[i]; idx_d[i] = frame_idx; }
    }
}
ORC_API void orc_base_fuse_f32(const float* bases /* N x npix x 3 */, size_t npix,
                               const int32_t* idx_e, const int32_t* idx_d, float* out) {
    for (size_t i = 0; i < npix; ++i)
        for (int c = 0; c < 3; ++c) {
            float a = bases[((size_t)idx_e[i] * npix + i) * 3 + c];
            float b = bases[((size_t)idx_d[i] * npix + i) * 3 + c];
            /* zeros + a + b with the np.where(.., img, 0) summation order; /2 exact */
            float lo = (idx_e[i] <= idx_d[i]) ? a : b, hi = (idx_e[i] <= idx_d[i]) ? b : a;
            float s = 0.0f + lo;
            s = s + hi;
            out[i * 3 + c] = s / 2.0f;
        }
}

/* img.astype(float32), pyramid.py:126 (parallel, into a caller-owned buffer) */
ORC_API void orc_to_f32_u8(const uint8_t* src, size_t n, float* dst) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) dst[i] = (float)src[i];
}
ORC_API void orc_to_f32_u16(const uint16_t* src, size_t n, float* dst) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) dst[i] = (float)src[i];
}

/* ---- synthetic stack generator, SURVEY.md 8(d) config 2 -------------------- */
static inline uint32_t lowbias32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
/* rows [y0, y0+h) x columns [x0, x0+w) of frame f of the H x W generator (checks on a corner of a full-size frame) */
ORC_API void orc_synth_crop_u8(uint8_t* out, int H, int W, int f, int N, uint32_t seed, int y0, int x0, int h, int w) {
    (void)W;
#pragma omp parallel for schedule(static)
    for (int yy = 0; yy < h; ++yy)
        for (int xx = 0; xx < w; ++xx)
            for (int c = 0; c < 3; ++c) {
                const int y = y0 + yy, x = x0 + xx;
                uint32_t hsh = lowbias32(seed ^ ((uint32_t)f * 0x9E3779B1U) ^ ((uint32_t)y * 0x85EBCA77U) ^
                                         ((uint32_t)x * 0xC2B2AE3DU) ^ (uint32_t)c);
                int noise = (int)(hsh >> 24) - 128;
                int band = (int)(((int64_t)y * N) / H);
                int d = band - f; if (d < 0) d = -d;
                int amp = 64 >> (d < 6 ? d : 6);
                int base = ((3 * x + 5 * y + 17 * c) & 127) + 64;
                int v = base + ((noise * amp) >> 7);
                if (v < 0) v = 0;
                if (v > 255) v = 255;
                out[((size_t)yy * w + xx) * 3 + c] = (uint8_t)v;
            }
}

ORC_API void orc_synth_frame_u8(uint8_t* out, int H, int W, int f, int N, uint32_t seed) {
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x)
            for (int c = 0; c < 3; ++c) {
                uint32_t hsh = lowbias32(seed ^ ((uint32_t)f * 0x9E3779B1U) ^
                                         ((uint32_t)y * 0x85EBCA77U) ^
                                         ((uint32_t)x * 0xC2B2AE3DU) ^ (uint32_t)c);
                int noise = (int)(hsh >> 24) - 128;
                int band = (int)(((int64_t)y * N) / H);
                int d = band - f; if (d < 0) d = -d;
                int amp = 64 >> (d < 6 ? d : 6);
                int base = ((3 * x + 5 * y + 17 * c) & 127) + 64;
                int v = base + ((noise * amp) >> 7);
                if (v < 0) v = 0;
                if (v > 255) v = 255;
                out[((size_t)y * W + x) * 3 + c] = (uint8_t)v;
            }
}

