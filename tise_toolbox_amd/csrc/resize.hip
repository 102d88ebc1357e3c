// PIL-exact uint8 bilinear resize fused with ToTensor() and the InceptionV3 input affine.
//
// Replaces transforms.Resize((299, 299)) + transforms.ToTensor() (reference
// image_realism/FID/fid_score.py:208-213 -> Pillow Image.resize(BILINEAR), 8 bits per channel) and the
// per-channel affine of image_realism/FID/inception.py:120-124.  Pillow's algorithm (Resample.c):
// per output coordinate a window [xmin, xmin+cnt) of normalised triangle weights rounded to 22-bit
// integers; out = clip8((sum src*k + 2^21) >> 22); horizontal pass first, stored as uint8, then
// the vertical pass on that uint8 image.  Integer arithmetic => the kernel is bit-exact.
//
// One workgroup produces RT output rows of one image: it stages the source rows those output
// rows depend on in LDS (16-byte coalesced loads), runs the horizontal pass LDS->LDS, then the
// vertical pass LDS->registers, maps each byte through a 3x256 fp32 table (v/255 then the affine,
// tabulated on the host with the reference's own op order) and stores fp32 rows coalesced, either
// planar NCHW or interleaved NHWC (torch.channels_last storage).
//
// Bound: HBM.  Algorithmic bytes per image: h*w*3 read + oh*ow*3*4 written
// (256x256 -> 299x299: 196 608 + 1 072 812 = 1 269 420 B).
#include <math.h>
#include <string.h>
#include <list>
#include <mutex>
#include <unordered_map>
#include <vector>
#include "common.h"

#define PRECISION_BITS 22

namespace {

struct CoeffTable {
    int in_size, out_size, ksize;
    std::vector<int> bounds;   // out_size * 2 : (min, count)
    std::vector<int> kk;       // out_size * ksize
};

double bilinear_filter(double x) {
    if (x < 0.0) x = -x;
    return x < 1.0 ? 1.0 - x : 0.0;
}

// Resample.c bicubic_filter (a = -0.5, support 2): the filter of clip._transform's Resize(224, BICUBIC)
// (text_relevance/RP_coco.py:31,64 and positional_alignment/PA.py:30,34 through clip.load's preprocess)
double bicubic_filter(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

// Resample.c precompute_coeffs() + normalize_coeffs_8bpc(); filter 0 = BILINEAR (support 1), 1 = BICUBIC (support 2)
void precompute_coeffs(int in_size, int out_size, CoeffTable& t, int filter) {
    double scale = (double)in_size / out_size;
    double filterscale = scale < 1.0 ? 1.0 : scale;
    double support = (filter == 1 ? 2.0 : 1.0) * filterscale;
    int ksize = (int)ceil(support) * 2 + 1;
    t.in_size = in_size; t.out_size = out_size; t.ksize = ksize;
    t.bounds.assign((size_t)out_size * 2, 0);
    t.kk.assign((size_t)out_size * ksize, 0);
    std::vector<double> k(ksize);
    const double ss = 1.0 / filterscale;
    for (int xx = 0; xx < out_size; ++xx) {
        double center = (xx + 0.5) * scale;
        double ww = 0.0;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        for (int x = 0; x < xmax; ++x) {
            double w = filter == 1 ? bicubic_filter((x + xmin - center + 0.5) * ss) : bilinear_filter((x + xmin - center + 0.5) * ss);
            k[x] = w;
            ww += w;
        }
        for (int x = 0; x < xmax; ++x) {
            if (ww != 0.0) k[x] /= ww;
            double v = k[x];
            t.kk[(size_t)xx * ksize + x] = v < 0 ? (int)(-0.5 + v * (1 << PRECISION_BITS)) : (int)(0.5 + v * (1 << PRECISION_BITS));
        }
        t.bounds[xx * 2 + 0] = xmin;
        t.bounds[xx * 2 + 1] = xmax;
    }
}

struct DevPlan {
    int device;      // HIP device the table lives on (one process may drive several GPUs)
    int h, w, oh, ow, filter;
    int ksx, ksy;
    int rt;          // output rows per workgroup
    int span;        // max source rows any row tile needs
    int* dev;        // bounds_x[ow*2] | kx[ow*ksx] | bounds_y[oh*2] | ky[oh*ksy] | tile_y0[ntiles]
    size_t cap;      // ints allocated at dev
    int off_kx, off_by, off_ky, off_t0, ntiles;
};

// Plan cache: least-recently-used list + hash index, keyed by (device, h, w, oh, ow).  Object crops (O-FID / O-IS)
// arrive in thousands of distinct sizes: the cache holds 8192 plans (~12 KB each) and an evicted plan's device
// buffer is REUSED for the new plan when it is large enough, so steady state does no hipMalloc / hipFree (hipFree
// is a device-wide synchronisation).
struct PlanKey {
    int device, h, w, oh, ow, filter;
    bool operator==(const PlanKey& o) const { return device == o.device && h == o.h && w == o.w && oh == o.oh && ow == o.ow && filter == o.filter; }
};
struct PlanKeyHash {
    size_t operator()(const PlanKey& k) const {
        size_t x = (size_t)k.device;
        for (int v : {k.h, k.w, k.oh, k.ow, k.filter}) x = x * 1000003u ^ (size_t)v;
        return x;
    }
};
constexpr size_t PLAN_CACHE_CAP = 8192;
std::mutex g_plan_mu;
std::list<DevPlan> g_plans;                                                   // front = most recently used
std::unordered_map<PlanKey, std::list<DevPlan>::iterator, PlanKeyHash> g_plan_index;

int get_plan(int h, int w, int oh, int ow, int filter, DevPlan* out) {
    int device = 0;
    TISE_HIP_CHECK(hipGetDevice(&device));
    std::lock_guard<std::mutex> lk(g_plan_mu);
    const PlanKey key{device, h, w, oh, ow, filter};
    auto hit = g_plan_index.find(key);
    if (hit != g_plan_index.end()) {
        g_plans.splice(g_plans.begin(), g_plans, hit->second);
        *out = g_plans.front();
        return TISE_OK;
    }
    CoeffTable tx, ty;
    precompute_coeffs(w, ow, tx, filter);
    precompute_coeffs(h, oh, ty, filter);
    DevPlan p;
    p.device = device;
    p.h = h; p.w = w; p.oh = oh; p.ow = ow; p.filter = filter; p.ksx = tx.ksize; p.ksy = ty.ksize;
    // pick the row tile so staged source rows + horizontal-pass rows fit comfortably in LDS
    const size_t lds_budget = 144 * 1024;
    int rt = 16;
    int span = 0;
    for (;; rt >>= 1) {
        span = 0;
        for (int y0 = 0; y0 < oh; y0 += rt) {
            int y1 = (y0 + rt < oh) ? y0 + rt : oh;
            int lo = ty.bounds[y0 * 2], hi = 0;
            for (int y = y0; y < y1; ++y) {
                int e = ty.bounds[y * 2] + ty.bounds[y * 2 + 1];
                if (e > hi) hi = e;
                if (ty.bounds[y * 2] < lo) lo = ty.bounds[y * 2];
            }
            if (hi - lo > span) span = hi - lo;
        }
        size_t need = (size_t)span * ((size_t)((w * 3 + 15) & ~15) + (size_t)((ow * 3 + 3) & ~3)) + 3 * 256 * 4;
        if (need <= lds_budget) break;
        if (rt == 1) return TISE_ERR_UNSUPPORTED;
    }
    p.rt = rt; p.span = span;
    p.ntiles = (oh + rt - 1) / rt;
    std::vector<int> host;
    host.insert(host.end(), tx.bounds.begin(), tx.bounds.end());
    p.off_kx = (int)host.size();
    host.insert(host.end(), tx.kk.begin(), tx.kk.end());
    p.off_by = (int)host.size();
    host.insert(host.end(), ty.bounds.begin(), ty.bounds.end());
    p.off_ky = (int)host.size();
    host.insert(host.end(), ty.kk.begin(), ty.kk.end());
    p.off_t0 = (int)host.size();
    for (int t = 0; t < p.ntiles; ++t) {
        int lo = ty.bounds[(t * rt) * 2];
        int y1 = (t * rt + rt < oh) ? t * rt + rt : oh;
        for (int y = t * rt; y < y1; ++y) if (ty.bounds[y * 2] < lo) lo = ty.bounds[y * 2];
        host.push_back(lo);
    }
    p.dev = nullptr;
    p.cap = 0;
    if (g_plans.size() >= PLAN_CACHE_CAP) {
        // evict the least recently used plan; any kernel still reading its table was enqueued before this call on
        // the caller's stream(s) -- order the overwrite behind them with a device synchronisation only in this
        // (rare: > 8192 live sizes) case
        DevPlan victim = g_plans.back();
        g_plan_index.erase(PlanKey{victim.device, victim.h, victim.w, victim.oh, victim.ow, victim.filter});
        g_plans.pop_back();
        TISE_HIP_CHECK(hipDeviceSynchronize());
        if (victim.device == device && victim.cap >= host.size()) { p.dev = victim.dev; p.cap = victim.cap; }
        else (void)hipFree(victim.dev);
    }
    if (!p.dev) {
        p.cap = host.size() + 1024;                                          // slack so a later, larger plan can reuse it
        TISE_HIP_CHECK(hipMalloc((void**)&p.dev, p.cap * sizeof(int)));
    }
    TISE_HIP_CHECK(hipMemcpy(p.dev, host.data(), host.size() * sizeof(int), hipMemcpyHostToDevice));
    g_plans.push_front(p);
    g_plan_index[key] = g_plans.begin();
    *out = p;
    return TISE_OK;
}

// device copies of the 3x256 fp32 byte->value tables, keyed by content (a run uses one or two)
struct LutEntry { int device; float host[3 * 256]; float* dev; };
std::mutex g_lut_mu;
std::vector<LutEntry*> g_luts;

int get_lut(const float* lut, float** dev) {
    int device = 0;
    TISE_HIP_CHECK(hipGetDevice(&device));
    std::lock_guard<std::mutex> lk(g_lut_mu);
    for (LutEntry* e : g_luts)
        if (e->device == device && memcmp(e->host, lut, sizeof(e->host)) == 0) { *dev = e->dev; return TISE_OK; }
    LutEntry* e = new LutEntry;
    e->device = device;
    memcpy(e->host, lut, sizeof(e->host));
    e->dev = nullptr;
    hipError_t err = hipMalloc((void**)&e->dev, sizeof(e->host));
    if (err == hipSuccess) err = hipMemcpy(e->dev, e->host, sizeof(e->host), hipMemcpyHostToDevice);
    if (err != hipSuccess) { tise_set_last_hip_error((int)err); if (e->dev) hipFree(e->dev); delete e; return TISE_ERR_HIP; }
    g_luts.push_back(e);
    *dev = e->dev;
    return TISE_OK;
}

__device__ __forceinline__ uint8_t clip8(int v) {
    v >>= PRECISION_BITS;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// SMALLK: every horizontal window has <= 3 taps (any up-scale): the thread keeps its taps in registers.
// Thread t owns the row elements e = t, t+256, ... (e = ox*3 + c) for EVERY row of the tile, so the
// per-element table look-ups (window start, taps, table row) are done once per thread, not per pixel,
// and the inner loops contain no integer division.
#define RS_MAXE 8     // elements per thread per row kept in registers: supports ow*3 <= 2048
template <bool SMALLK>
__global__ __launch_bounds__(256) void resize_bilinear_u8_kernel(
    const uint8_t* __restrict__ src, int h, int w, float* __restrict__ dst, int oh, int ow, int nhwc,
    const int* __restrict__ plan, int ksx, int ksy, int off_kx, int off_by, int off_ky, int off_t0, int rt, int span,
    const float* __restrict__ lut, uint8_t* __restrict__ u8_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int src_pitch = (w * 3 + 15) & ~15;
    const int tmp_pitch = (ow * 3 + 3) & ~3;
    float* lut_s = reinterpret_cast<float*>(smem);                       // 3*256 floats
    unsigned char* srow = smem + 3 * 256 * 4;                            // span x src_pitch
    unsigned char* trow = srow + (size_t)span * src_pitch;               // span x tmp_pitch
    const int* bx = plan;
    const int* kx = plan + off_kx;
    const int* by = plan + off_by;
    const int* ky = plan + off_ky;

    const int img = blockIdx.y;
    const int tile = blockIdx.x;
    const int oy0 = tile * rt;
    const int oy1 = min(oy0 + rt, oh);
    const int ys0 = plan[off_t0 + tile];
    int ys1 = 0;                                                         // last source row needed by this tile
    for (int y = oy0; y < oy1; ++y) ys1 = max(ys1, by[2 * y] + by[2 * y + 1]);
    const int nrows = ys1 - ys0;
    const int tid = threadIdx.x;
    const int rowlen = ow * 3;

    for (int i = tid; i < 3 * 256; i += 256) lut_s[i] = lut[i];

    // stage source rows [ys0, ys1): rows are contiguous in memory (w*3 bytes each)
    const uint8_t* sbase = src + ((size_t)img * h + ys0) * (size_t)w * 3;
    const int row_bytes = w * 3;
    if ((((uintptr_t)sbase) & 15) == 0 && (row_bytes & 15) == 0) {
        const int vec_per_row = row_bytes >> 4;
        for (int i = tid; i < nrows * vec_per_row; i += 256) {
            const int r = i / vec_per_row, c = i - r * vec_per_row;
            *reinterpret_cast<uint4*>(srow + (size_t)r * src_pitch + c * 16) =
                *reinterpret_cast<const uint4*>(sbase + (size_t)r * row_bytes + c * 16);
        }
    } else {
        const size_t tot = (size_t)nrows * row_bytes;
        for (size_t i = tid; i < tot; i += 256) {
            const int r = (int)(i / row_bytes), c = (int)(i - (size_t)r * row_bytes);
            srow[(size_t)r * src_pitch + c] = sbase[i];
        }
    }

    // per-thread element descriptors
    int e_off[RS_MAXE];          // byte offset of the first tap inside a staged source row
    int e_k[RS_MAXE][3];         // taps (SMALLK)
    int e_cnt[RS_MAXE];
    int e_lut[RS_MAXE];          // c * 256
    int e_dst[RS_MAXE];          // offset of (ox, c) inside an output row (nhwc) / inside a plane row (nchw: ox)
    int ne = 0;
#pragma unroll
    for (int j = 0; j < RS_MAXE; ++j) {
        const int e = tid + 256 * j;
        e_off[j] = 0; e_cnt[j] = 0; e_lut[j] = 0; e_dst[j] = 0;
        e_k[j][0] = e_k[j][1] = e_k[j][2] = 0;
        if (e < rowlen) {
            ne = j + 1;
            const int ox = e / 3, c = e - 3 * ox;
            const int xmin = (w == ow) ? ox : bx[2 * ox];
            e_off[j] = xmin * 3 + c;
            e_cnt[j] = (w == ow) ? 1 : bx[2 * ox + 1];
            e_lut[j] = c * 256;
            e_dst[j] = nhwc ? e : ox;
            if (SMALLK && w != ow) {
                const int* k = kx + ox * ksx;
                e_k[j][0] = k[0];
                e_k[j][1] = e_cnt[j] > 1 ? k[1] : 0;
                e_k[j][2] = e_cnt[j] > 2 ? k[2] : 0;
            }
        }
    }
    __syncthreads();

    // horizontal pass LDS -> LDS (uint8 intermediate, as Pillow).  When w == ow Pillow skips this pass.
    for (int r = 0; r < nrows; ++r) {
        const unsigned char* sr = srow + (size_t)r * src_pitch;
        unsigned char* tr = trow + (size_t)r * tmp_pitch;
#pragma unroll
        for (int j = 0; j < RS_MAXE; ++j) {
            if (j >= ne) break;
            const int e = tid + 256 * j;
            if (e >= rowlen) break;
            const unsigned char* sp = sr + e_off[j];
            if (w == ow) {
                tr[e] = sp[0];
            } else if (SMALLK) {
                // taps beyond cnt carry a zero coefficient; clamp their address to stay inside the staged row
                const int c1 = e_cnt[j] > 1 ? 3 : 0, c2 = e_cnt[j] > 2 ? 6 : 0;
                // 24-bit multiplies (a byte x a weight <= 2^22): full-rate v_mad_u32_u24 instead of quarter-rate 32-bit multiplies
                const int ss = (1 << (PRECISION_BITS - 1)) + __mul24((int)sp[0], e_k[j][0]) + __mul24((int)sp[c1], e_k[j][1]) +
                               __mul24((int)sp[c2], e_k[j][2]);
                tr[e] = clip8(ss);
            } else {
                const int ox = e / 3;
                const int* k = kx + ox * ksx;
                int ss = 1 << (PRECISION_BITS - 1);
                for (int x = 0; x < e_cnt[j]; ++x) ss += __mul24((int)sp[x * 3], k[x]);
                tr[e] = clip8(ss);
            }
        }
    }
    __syncthreads();

    // uint8-only output (the product path: the stem convolution applies the table) with <= 3 vertical taps: FOUR output bytes
    // per lane -- three aligned dword reads of the horizontal-pass rows instead of twelve byte reads, one (unaligned: a row is
    // ow * 3 bytes) dword store instead of four byte stores.  Same integer arithmetic per byte.
    if (!dst && u8_out && h != oh && ksy <= 3) {
        const int nd = (rowlen + 3) >> 2;
        for (int y = oy0; y < oy1; ++y) {
            const int ymin = by[2 * y], cnt = by[2 * y + 1];
            const int* k = ky + y * ksy;
            const int k0 = k[0], k1 = cnt > 1 ? k[1] : 0, k2 = cnt > 2 ? k[2] : 0;
            const unsigned char* tbase = trow + (size_t)(ymin - ys0) * tmp_pitch;
            const int p1 = cnt > 1 ? tmp_pitch : 0, p2 = cnt > 2 ? 2 * tmp_pitch : 0;
            uint8_t* urow = u8_out + ((size_t)img * oh + y) * (size_t)rowlen;
            for (int dw = tid; dw < nd; dw += 256) {
                const unsigned t0 = *reinterpret_cast<const unsigned*>(tbase + 4 * dw);
                const unsigned t1 = *reinterpret_cast<const unsigned*>(tbase + p1 + 4 * dw);
                const unsigned t2 = *reinterpret_cast<const unsigned*>(tbase + p2 + 4 * dw);
                unsigned outw = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int ss = (1 << (PRECISION_BITS - 1)) + __mul24((int)((t0 >> (8 * b)) & 255u), k0) +
                                   __mul24((int)((t1 >> (8 * b)) & 255u), k1) + __mul24((int)((t2 >> (8 * b)) & 255u), k2);
                    outw |= (unsigned)clip8(ss) << (8 * b);
                }
                if (4 * dw + 3 < rowlen) {
                    __builtin_memcpy(urow + 4 * dw, &outw, 4);
                } else {
                    for (int b = 0; b < 4; ++b)
                        if (4 * dw + b < rowlen) urow[4 * dw + b] = (uint8_t)(outw >> (8 * b));
                }
            }
        }
        return;
    }

    // vertical pass + table + store, one output row at a time (row parameters are wave-uniform)
    for (int y = oy0; y < oy1; ++y) {
        const int ymin = (h == oh) ? y : by[2 * y];
        const int cnt = (h == oh) ? 1 : by[2 * y + 1];
        const int* k = ky + y * ksy;
        const unsigned char* tbase = trow + (size_t)(ymin - ys0) * tmp_pitch;
        int k0 = 0, k1 = 0, k2 = 0;
        if (h != oh && cnt <= 3) { k0 = k[0]; k1 = cnt > 1 ? k[1] : 0; k2 = cnt > 2 ? k[2] : 0; }
        const int p1 = cnt > 1 ? tmp_pitch : 0, p2 = cnt > 2 ? 2 * tmp_pitch : 0;
        float* drow_nhwc = dst ? dst + ((size_t)img * oh + y) * (size_t)rowlen : nullptr;
        uint8_t* urow = u8_out ? u8_out + ((size_t)img * oh + y) * (size_t)rowlen : nullptr;
#pragma unroll
        for (int j = 0; j < RS_MAXE; ++j) {
            if (j >= ne) break;
            const int e = tid + 256 * j;
            if (e >= rowlen) break;
            const unsigned char* tp = tbase + e;
            uint8_t v;
            if (h == oh) {
                v = tp[0];
            } else if (cnt <= 3) {
                const int ss = (1 << (PRECISION_BITS - 1)) + __mul24((int)tp[0], k0) + __mul24((int)tp[p1], k1) + __mul24((int)tp[p2], k2);
                v = clip8(ss);
            } else {
                int ss = 1 << (PRECISION_BITS - 1);
                for (int yy = 0; yy < cnt; ++yy) ss += __mul24((int)tp[(size_t)yy * tmp_pitch], k[yy]);
                v = clip8(ss);
            }
            if (dst) {                                      // dst == nullptr: uint8 result only (the stem conv applies the table)
                const float f = lut_s[e_lut[j] + v];
                if (nhwc) drow_nhwc[e] = f;
                else dst[(((size_t)img * 3 + (e_lut[j] >> 8)) * oh + y) * ow + e_dst[j]] = f;
            }
            if (urow) urow[e] = v;
        }
    }
}

}  // namespace

static int resize_u8_impl(const uint8_t* src_dev, int n, int h, int w, float* dst_dev, int oh, int ow,
                          int nhwc, const float* lut, uint8_t* u8_out_dev, int filter, void* stream) {
    if (n < 0 || h <= 0 || w <= 0 || oh <= 0 || ow <= 0 || !lut || (n > 0 && (!src_dev || (!dst_dev && !u8_out_dev))) || (filter != 0 && filter != 1))
        return TISE_ERR_INVALID_ARG;
    if (n == 0) return TISE_OK;
    if (n > 65535) return TISE_ERR_UNSUPPORTED;
    DevPlan p;
    int rc = get_plan(h, w, oh, ow, filter, &p);
    if (rc != TISE_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    float* lut_dev = nullptr;
    rc = get_lut(lut, &lut_dev);
    if (rc != TISE_OK) return rc;
    const size_t lds = 3 * 256 * 4 + (size_t)p.span * (((w * 3 + 15) & ~15) + ((ow * 3 + 3) & ~3));
    if (ow * 3 > 256 * RS_MAXE) return TISE_ERR_UNSUPPORTED;
    if (p.ksx <= 3) {
        if (lds > 48 * 1024)
            TISE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(resize_bilinear_u8_kernel<true>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(resize_bilinear_u8_kernel<true>, dim3(p.ntiles, n), dim3(256), lds, st, src_dev, h, w, dst_dev,
                           oh, ow, nhwc, p.dev, p.ksx, p.ksy, p.off_kx, p.off_by, p.off_ky, p.off_t0, p.rt, p.span,
                           lut_dev, u8_out_dev);
    } else {
        if (lds > 48 * 1024)
            TISE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(resize_bilinear_u8_kernel<false>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(resize_bilinear_u8_kernel<false>, dim3(p.ntiles, n), dim3(256), lds, st, src_dev, h, w, dst_dev,
                           oh, ow, nhwc, p.dev, p.ksx, p.ksy, p.off_kx, p.off_by, p.off_ky, p.off_t0, p.rt, p.span,
                           lut_dev, u8_out_dev);
    }
    TISE_LAUNCH_CHECK();
    return TISE_OK;
}

extern "C" int tise_resize_bilinear_u8(const uint8_t* src_dev, int n, int h, int w, float* dst_dev, int oh, int ow,
                                       int nhwc, const float* lut, uint8_t* u8_out_dev, void* stream) {
    return resize_u8_impl(src_dev, n, h, w, dst_dev, oh, ow, nhwc, lut, u8_out_dev, 0, stream);
}

// The same two-pass 8-bit resample with Pillow's BICUBIC filter (filter = 1; 0 = BILINEAR): clip._transform's
// Resize(224, interpolation=BICUBIC) + ToTensor + Normalize through the table (RP_coco.py:64, PA.py:34).
extern "C" int tise_resize_u8(const uint8_t* src_dev, int n, int h, int w, float* dst_dev, int oh, int ow, int nhwc,
                              const float* lut, uint8_t* u8_out_dev, int filter, void* stream) {
    return resize_u8_impl(src_dev, n, h, w, dst_dev, oh, ow, nhwc, lut, u8_out_dev, filter, stream);
}
