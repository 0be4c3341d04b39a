// k_crop.hip -- PointCloud2 row ingest fused with the crop box and its
// order-preserving compaction; also emits the search-grid cell key of every
// survivor so no later pass has to re-read the cloud for it.
//
// Replaces pcl::fromROSMsg (/root/reference src/geometric_mapping.cpp:55) and
// chopCloud -> pcl::CropBox (/root/reference src/tunnel_processing.cpp:39-49).
// HBM-bound: one launch (chained scan, gm_compact.hpp) reads point_step bytes per
// input row -- the survivors' rows a second time out of L2 when they are written --
// and writes 16 + 4 bytes per survivor.
#include "gm_compact.hpp"
#include "gm_internal.hpp"

namespace gm {

__device__ __forceinline__ float load_f32_bytes(const uint8_t *p)
{
    uint32_t u = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
    return __uint_as_float(u);
}

struct RowReader {
    RowLayout L;
    __device__ __forceinline__ void load(uint32_t i, float &x, float &y, float &z) const
    {
        const uint8_t *row = (L.data_at ? *L.data_at : L.data) + (size_t)i * L.step;
        if (L.mode == 0) {
            const float4 v = *reinterpret_cast<const float4 *>(row);   // (a non-temporal load here: no change, 91.2 vs 90.7 us at 10 M points)
            x = v.x; y = v.y; z = v.z;
        } else if (L.mode == 1) {
            x = *reinterpret_cast<const float *>(row + L.ox);
            y = *reinterpret_cast<const float *>(row + L.oy);
            z = *reinterpret_cast<const float *>(row + L.oz);
        } else {
            x = load_f32_bytes(row + L.ox);
            y = load_f32_bytes(row + L.oy);
            z = load_f32_bytes(row + L.oz);
        }
        if (L.bswap) {
            x = __uint_as_float(__builtin_bswap32(__float_as_uint(x)));
            y = __uint_as_float(__builtin_bswap32(__float_as_uint(y)));
            z = __uint_as_float(__builtin_bswap32(__float_as_uint(z)));
        }
    }
};

// digit counters of the block: [pass][bin] for the passes of the cell sort (k_sort.hip radix_plan)
__device__ __forceinline__ uint32_t *crop_digit_hist()
{
    __shared__ uint32_t h[4 << 10];
    return h;
}

// CropBox::applyFilter: outside iff any coordinate < min or > max (closed box);
// NaN rows dropped (see gm_hip.h / DESIGN.md for the dense-flag caveat).
struct CropPred {
    RowReader rd;
    float lo, hi;
    GridParams g;
    // The digit totals of the cell sort's first pass do not depend on the order of the keys, and the key of a survivor
    // is known right here: the block counts them in LDS (zeroed by CropEmit::prepare, added to the totals the pass starts
    // from by CropEmit::finish) -- in THIS phase, in front of the chained scan's look-back, where the block would wait
    // anyway (counting in the emit step, behind the look-back, cost the 10 M-point crop 8 us).  count_bits == 0: not asked for.
    uint32_t count_bits;
    struct Payload { float x, y, z; uint32_t key; };   // the row's coordinates and cell key: the emit step writes them, the rows are read once
    __device__ __forceinline__ bool operator()(uint32_t i, Payload &p) const
    {
        rd.load(i, p.x, p.y, p.z);
        if (!finite3(p.x, p.y, p.z)) return false;
        if (p.x < lo || p.y < lo || p.z < lo || p.x > hi || p.y > hi || p.z > hi) return false;
        p.key = cell_key(g, p.x, p.y, p.z);
        if (count_bits) atomicAdd(&crop_digit_hist()[p.key & ((1u << count_bits) - 1u)], 1u);
        return true;
    }
};

struct CropEmit {
    static constexpr bool kHasFinish = true, kHasPrepare = true;
    RowReader rd;
    GridParams g;
    float4 *__restrict__ crop4;
    uint32_t *__restrict__ keys;
    // the block's digit counts (CropPred) -> the totals the cell sort's first pass starts from; plan.passes == 0: not asked for
    SortPlan plan;
    uint32_t *__restrict__ totals;   // [pass][2048]
    __device__ __forceinline__ void prepare() const
    {
        if (plan.passes) {
            uint32_t *h = crop_digit_hist();
            for (uint32_t k = threadIdx.x; k < ((uint32_t)plan.passes << plan.bits); k += blockDim.x) h[k] = 0u;
        }
    }
    __device__ __forceinline__ void finish(uint32_t) const
    {
        if (plan.passes) {
            __syncthreads();   // every emit of the block has counted
            uint32_t *h = crop_digit_hist();
            const uint32_t bins = 1u << plan.bits;
            for (uint32_t k = threadIdx.x; k < ((uint32_t)plan.passes << plan.bits); k += blockDim.x) {
                const uint32_t c = h[k];
                if (c) atomicAdd(&totals[(k >> plan.bits) * 2048u + (k & (bins - 1u))], c);
            }
        }
    }
    // (The keys leave as one dword per lane, 256 bytes per store instruction; tools/microbench/stream_probe.hip prices those 33 MB
    // of the 10 M-point frame at 14 us, as much as 80 MB of 16-byte-per-lane stores.  Collecting a wave's keys in LDS --
    // its survivors are consecutive in the output -- and writing them 16 bytes per lane was measured: 80 -> 86 us, the
    // stores then wait for the wave's last item.)
    __device__ __forceinline__ void operator()(uint32_t src, uint32_t dst, const CropPred::Payload &p) const
    {
        crop4[dst] = make_float4(p.x, p.y, p.z, __uint_as_float(src));
        keys[dst] = p.key;
    }
};

__global__ __launch_bounds__(256) void k_zero_fill(ZeroJobs jobs)
{
    const uint2 z = make_uint2(0u, 0u);
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        uint2 *p = reinterpret_cast<uint2 *>(jobs.ptr[j]);
        const uint64_t nw = jobs.words8[j];
        for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nw; i += (uint64_t)gridDim.x * blockDim.x) p[i] = z;
    }
    // one frame more on this slot (the chained scans of a replayed graph take their epochs from this count)
    if (jobs.frame_counter && blockIdx.x == 0 && threadIdx.x == 0) *jobs.frame_counter += 1u;
}

void launch_zero_fill(const ZeroJobs &jobs, hipStream_t s)
{
    uint64_t most = 0;
    for (int j = 0; j < 6; ++j) most = jobs.words8[j] > most ? jobs.words8[j] : most;
    uint32_t nb = (uint32_t)((most + 255) / 256);
    if (nb > 1024u) nb = 1024u;
    if (nb == 0) nb = 1;
    hipLaunchKernelGGL(k_zero_fill, dim3(nb), dim3(256), 0, s, jobs);
}

// n_size >= n sizes the launch; the point count is n, or *n_dev when given (a captured launch: its arguments are frozen)
void launch_crop(const RowLayout &rows, uint32_t n, float lo, float hi, const GridParams &g, Slot &sl, hipStream_t s,
                 uint32_t n_size, const uint32_t *n_dev, bool count_digits)
{
    if (n_size < n) n_size = n;
    if (n_size == 0) return;  // counters were zeroed: n_cropped stays 0
    RowReader rd{rows};
    CropPred pred{rd, lo, hi, g, count_digits ? (uint32_t)radix_plan(cell_key_bits(g)).bits : 0u};
    // (the digit totals of the cell sort's FIRST pass only: the pass itself counts the later ones, k_sort.hip)
    CropEmit emit{rd, g, sl.crop4, sl.keys_a, count_digits ? SortPlan{1, radix_plan(cell_key_bits(g)).bits} : SortPlan{0, 0}, sl.sort.totals};
    // Tile shape: 512 threads x 8 rows, and 1024 x 8 for frames beyond 2 M points.  Every tile costs a ticket and a
    // look-back; with every block slot of the chip taken (2.4 rounds of 4096-point tiles at 10 M points) twice the tile
    // is 107 -> 80 us on the 10 M-point frame and 34.5 -> 31 us at 3 M, while the 1 M-point frame (204 tiles for 256 CUs)
    // wants the small one (17.5 against 21 us for 512 x 16).  Measured around it at 10 M points: 512x16 89 us, 512x24
    // 86, 512x32 116, 1024x12 85, 1024x16 91, 256x32 100, 512x4 and 256x8 143; MORE resident blocks per CU are slower
    // (1024x8 held to 64 registers, two blocks per CU: 103 us; 512x16 at three per CU: 91), fewer too (512x8, one per
    // CU: 131).  What the pattern allows without any scan: tools/microbench/stream_probe.hip, 66-74 us.
    // GM_CROP_TILE=<threads>x<items>: experiments.
    static const char *e = getenv("GM_CROP_TILE");
    int th = 512, it = 8;
    if (e) sscanf(e, "%dx%d", &th, &it);
    else if (n_size > 2000000u) th = 1024;
    const ScanState st = next_scan(sl);
#define GM_CROP_LAUNCH(T, I)                                                                                              \
    hipLaunchKernelGGL((k_compact<CropPred, CropEmit, T, I>), dim3((n_size + (T) * (I) - 1u) / ((T) * (I))), dim3(T), 0, s, pred, \
                       emit, n_dev, n, st, &sl.ctr->n_cropped, (uint32_t *)nullptr)
    if (th == 1024 && it == 8) GM_CROP_LAUNCH(1024, 8);
    else if (th == 512 && it == 16) GM_CROP_LAUNCH(512, 16);
    else GM_CROP_LAUNCH(512, 8);
#undef GM_CROP_LAUNCH
}

}  // namespace gm
