// (a2) PNG row filters on the device: the second half of ``Image.open(f).convert("RGB")`` of the reference's
// Dataset.__getitem__ (image_realism/FID/img_data.py:19-25; third-party Pillow -> zlib inflate + the five row filters of
// RFC 2083 section 6).  The decode processes of the image feed only INFLATE a file (csrc/png_decode.c:
// tise_png_inflate_slot) into a slot of the shared page-locked ring; the slots travel to HBM as they are and this kernel
// reverses the filters and drops the alpha byte, writing the (n, h, w, 3) uint8 batch the resize kernel reads.
//
// Integer, byte-exact work with a dependency chain: a reconstructed byte needs its left (a), upper (b) and upper-left (c)
// neighbours of the same channel (Sub: a, Up: b, Average: (a + b) >> 1, Paeth: the one of a, b, c nearest to a + b - c).
// ONE WAVE PER IMAGE walks the image in blocks of up to 64 rows on a skewed front: lane j owns row y0 + j and is one
// pixel behind lane j - 1, so at every step its upper neighbour is what the lane above produced one step earlier -- it
// arrives by DPP (wave_shr:1, a VALU move, no LDS round trip), the upper-left neighbour is last step's upper one, the left
// one the lane's own last output.  A block takes w + rows - 1 steps; 256 x 256: 4 blocks x 319 steps of ~70 instructions.
// The filtered rows of a block are staged in LDS (coalesced dword loads + v_alignbyte for the 1-byte filter prefix that
// misaligns every row), reconstructed IN PLACE (RGBA rows compact to RGB as they go: byte 3x of a row is written after
// byte 4x was read), and leave as whole dwords.  The LDS row pitch is an odd number of dwords: the 64 lanes' byte
// accesses fall on different banks.  Channels are independent chains, so alpha is never reconstructed.
//
// Bound: neither HBM (3000 images: 1.2 GB in + out, 0.15 ms at 8 TB/s) nor any throughput -- the chain: ~1300 steps per
// image, three waves per SIMD interleaved.  It runs on the feed's side stream beside the trunk.
#include "common.h"

namespace {

constexpr int HDR = 64;                 // slot header bytes (png_decode.c: TISE_PNG_SLOT_HDR)
constexpr int LDS_BYTES = 64 * 1024;    // static limit of a workgroup without an attribute; three workgroups fit a CU
constexpr int ROW_MAX = 8192;           // png_decode.c: TISE_PNG_DEVICE_ROW_MAX

__device__ __forceinline__ int wave_shr1(int old, int v) {
    // lane l receives lane l-1's v; lane 0 keeps `old` (DPP wave_shr:1, bound_ctrl off)
    return __builtin_amdgcn_update_dpp(old, v, 0x138, 0xf, 0xf, false);
}

// |x - y| for unsigned values (v_sad_u32)
__device__ __forceinline__ unsigned absdiff(unsigned x, unsigned y) {
    unsigned r;
    asm("v_sad_u32 %0, %1, %2, 0" : "=v"(r) : "v"(x), "v"(y));
    return r;
}

__device__ __forceinline__ unsigned predict(unsigned a, unsigned b, unsigned c, bool f1, bool f2, bool f3, bool f4) {
    const unsigned p = a + b;
    const unsigned pa = absdiff(b, c), pb = absdiff(a, c), pc = absdiff(p, c << 1);
    const unsigned bc = pb <= pc ? b : c;
    const unsigned paeth = pa <= min(pb, pc) ? a : bc;
    unsigned r = f1 ? a : 0u;
    r = f2 ? b : r;
    r = f3 ? (p >> 1) : r;
    r = f4 ? paeth : r;
    return r;
}

template <int BPP>
__device__ void unfilter_image(const uint8_t* __restrict__ pay, uint8_t* __restrict__ dst, int h, int w, uint8_t* lds,
                               int pitch, int rows_per_block) {
    const int lane = threadIdx.x;
    const int rb = w * BPP + 1;                       // bytes of a filtered row in the payload
    const int out_row = w * 3;
    const int row_dwords = (w * BPP + 3) >> 2;
    const int nslot = rows_per_block + 1;             // LDS row slots: the block's rows + the row above the block
    int pslot = 0;                                    // slot of the row above the block (zeros above the image)
    for (int i = lane; i < (pitch >> 2); i += 64) reinterpret_cast<uint32_t*>(lds)[i] = 0u;
    const bool dword_out = ((out_row & 3) == 0) && ((reinterpret_cast<uintptr_t>(dst) & 3) == 0);
    for (int y0 = 0; y0 < h; y0 += rows_per_block) {
        const int rows = min(rows_per_block, h - y0);
        // row r of the block lives in slot (pslot + 1 + r) mod nslot: a full block's last row is then already where the next
        // block looks for its upper row (slot pslot - 1), nothing is copied between blocks
        // ---- stage: rows y0 .. y0+rows-1, filter byte stripped ----
        for (int r = 0; r < rows; ++r) {
            const size_t s = (size_t)(y0 + r) * rb + 1;
            const uint32_t* src = reinterpret_cast<const uint32_t*>(pay + (s & ~(size_t)3));
            const int sh = (int)(s & 3);
            int sl = pslot + 1 + r; sl = sl >= nslot ? sl - nslot : sl;
            uint32_t* drow = reinterpret_cast<uint32_t*>(lds + sl * pitch);
            for (int i = lane; i < row_dwords; i += 64) drow[i] = __builtin_amdgcn_alignbyte(src[i + 1], src[i], sh);
        }
        int ft = 0;
        if (lane < rows) ft = pay[(size_t)(y0 + lane) * rb];
        const bool f1 = ft == 1, f2 = ft == 2, f3 = ft == 3, f4 = ft == 4;
        __syncthreads();
        // ---- the skewed front ----
        const int lrow = min(lane, rows - 1);                    // lanes beyond the block's rows shadow its last row (reads only)
        int msl = pslot + 1 + lrow; msl = msl >= nslot ? msl - nslot : msl;
        uint8_t* mine = lds + msl * pitch;
        const uint8_t* prev = lds + pslot * pitch;
        unsigned a0 = 0, a1 = 0, a2 = 0, c0 = 0, c1 = 0, c2 = 0;
        const int steps = w + rows - 1;
        // Software-pipelined by hand and branch-free: the LDS reads of step t + 1 (the lane's filtered pixel; the pixel of the
        // row above the block, which only lane 0 uses) are issued before the arithmetic of step t at clamped addresses by every
        // lane, so no step waits on an LDS round trip and the loop body is one basic block; only the three stores are masked.
        unsigned gn0 = mine[0], gn1 = mine[1], gn2 = mine[2];
        int pn0 = prev[0], pn1 = prev[1], pn2 = prev[2];
        for (int t = 0; t < steps; ++t) {
            const int x = t - lane;
            const bool active = lane < rows && (unsigned)x < (unsigned)w;
            const int p0 = pn0, p1 = pn1, p2 = pn2;
            const unsigned g0 = gn0, g1 = gn1, g2 = gn2;
            {
                const int xn = BPP * min(max(x + 1, 0), w - 1), tn = 3 * min(t + 1, w - 1);
                gn0 = mine[xn]; gn1 = mine[xn + 1]; gn2 = mine[xn + 2];
                pn0 = prev[tn]; pn1 = prev[tn + 1]; pn2 = prev[tn + 2];
            }
            // upper neighbour: lane 0 has read the row above the block, the others get the lane above's last output
            const unsigned b0 = (unsigned)wave_shr1(p0, (int)a0), b1 = (unsigned)wave_shr1(p1, (int)a1), b2 = (unsigned)wave_shr1(p2, (int)a2);
            const unsigned n0 = (g0 + predict(a0, b0, c0, f1, f2, f3, f4)) & 255u;
            const unsigned n1 = (g1 + predict(a1, b1, c1, f1, f2, f3, f4)) & 255u;
            const unsigned n2 = (g2 + predict(a2, b2, c2, f1, f2, f3, f4)) & 255u;
            if (active) { mine[3 * x] = (uint8_t)n0; mine[3 * x + 1] = (uint8_t)n1; mine[3 * x + 2] = (uint8_t)n2; }
            // next step: upper-left = this step's upper, left = this step's output; a lane outside its row holds zeros
            c0 = active ? b0 : 0u; c1 = active ? b1 : 0u; c2 = active ? b2 : 0u;
            a0 = active ? n0 : 0u; a1 = active ? n1 : 0u; a2 = active ? n2 : 0u;
        }
        __syncthreads();
        // ---- write the block out: RGB rows, contiguous in dst ----
        for (int r = 0; r < rows; ++r) {
            int sl = pslot + 1 + r; sl = sl >= nslot ? sl - nslot : sl;
            uint8_t* orow = dst + (size_t)(y0 + r) * out_row;
            if (dword_out) {
                const uint32_t* srow = reinterpret_cast<const uint32_t*>(lds + sl * pitch);
                for (int i = lane; i < (out_row >> 2); i += 64) reinterpret_cast<uint32_t*>(orow)[i] = srow[i];
            } else {
                const uint8_t* srow = lds + sl * pitch;
                for (int i = lane; i < out_row; i += 64) orow[i] = srow[i];
            }
        }
        pslot += rows; pslot = pslot >= nslot ? pslot - nslot : pslot;      // the block's last row
        __syncthreads();
    }
}

__global__ __launch_bounds__(64) void png_unfilter_kernel(const uint8_t* __restrict__ slots, int64_t slot_stride, int h, int w,
                                                           uint8_t* __restrict__ dst, int pitch3, int rows3, int pitch4, int rows4) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int64_t img = blockIdx.x;
    const uint8_t* slot = slots + img * slot_stride;
    uint8_t* out = dst + img * (int64_t)h * w * 3;
    const int mode = slot[0];
    const uint8_t* pay = slot + HDR;
    if (mode == 3) {
        unfilter_image<3>(pay, out, h, w, lds, pitch3, rows3);
    } else if (mode == 4 && rows4 > 0) {
        unfilter_image<4>(pay, out, h, w, lds, pitch4, rows4);
    } else {
        // pixels decoded on the host: copy (dwords when both sides allow it)
        const int64_t nbytes = (int64_t)h * w * 3;
        if ((reinterpret_cast<uintptr_t>(out) & 3) == 0) {
            const int64_t nd = nbytes >> 2;
            for (int64_t i = threadIdx.x; i < nd; i += 64) reinterpret_cast<uint32_t*>(out)[i] = reinterpret_cast<const uint32_t*>(pay)[i];
            for (int64_t i = (nd << 2) + threadIdx.x; i < nbytes; i += 64) out[i] = pay[i];
        } else {
            for (int64_t i = threadIdx.x; i < nbytes; i += 64) out[i] = pay[i];
        }
    }
}

// LDS row pitch for rows of `bytes` data bytes: whole dwords, an ODD number of them (lane j's byte of column x sits
// (j * pitch + x) / 4 dwords in: an odd pitch spreads the lanes of a 32-lane group over all 32 banks).
int lds_pitch(int bytes) {
    int d = (bytes + 3) / 4 + 1;             // + 1: the staging loop writes whole dwords and reads one dword ahead
    if ((d & 1) == 0) ++d;
    return d * 4;
}

}  // namespace

extern "C" int tise_png_unfilter_rgb8(const uint8_t* slots_dev, int64_t n, int64_t slot_stride, int h, int w, uint8_t* dst_dev,
                                      void* stream) {
    if (n < 0 || h <= 0 || w <= 0 || slot_stride < HDR) return TISE_ERR_INVALID_ARG;
    if (n == 0) return TISE_OK;
    if (!slots_dev || !dst_dev) return TISE_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(slots_dev) & 3) || (slot_stride & 3)) return TISE_ERR_INVALID_ARG;   // dword staging
    if (n > 0x7fffffff) return TISE_ERR_UNSUPPORTED;
    if (slot_stride < HDR + (int64_t)h * w * 3) return TISE_ERR_INVALID_ARG;
    // Which filtered forms can these slots hold at all (png_decode.c: tise_png_slot_bytes / tise_png_inflate_slot write a
    // filtered payload only when it fits with 8 bytes of slack and its rows fit ROW_MAX)?  Rows per block: what fits the
    // LDS tile beside the upper row, at most one per lane.
    const bool can3 = slot_stride >= HDR + (int64_t)h * ((int64_t)w * 3 + 1) + 8 && (int64_t)w * 3 + 1 <= ROW_MAX;
    const bool can4 = slot_stride >= HDR + (int64_t)h * ((int64_t)w * 4 + 1) + 8 && (int64_t)w * 4 + 1 <= ROW_MAX;
    int p3 = 0, p4 = 0, r3 = 0, r4 = 0;
    size_t lds = 0;
    if (can3) {
        p3 = lds_pitch(w * 3);
        r3 = LDS_BYTES / p3 - 1;
        r3 = r3 > 64 ? 64 : r3;
        r3 = r3 > h ? h : r3;
        lds = (size_t)p3 * (r3 + 1);
    }
    if (can4) {
        p4 = lds_pitch(w * 4);
        r4 = LDS_BYTES / p4 - 1;
        r4 = r4 > 64 ? 64 : r4;
        r4 = r4 > h ? h : r4;
        const size_t l4 = (size_t)p4 * (r4 + 1);
        lds = l4 > lds ? l4 : lds;
    }
    if ((can3 && r3 < 1) || (can4 && r4 < 1)) return TISE_ERR_UNSUPPORTED;       // cannot happen below ROW_MAX
    hipLaunchKernelGGL(png_unfilter_kernel, dim3((unsigned)n), dim3(64), lds, (hipStream_t)stream, slots_dev, slot_stride, h, w,
                       dst_dev, p3, r3, p4, r4);
    TISE_LAUNCH_CHECK();
    return TISE_OK;
}
