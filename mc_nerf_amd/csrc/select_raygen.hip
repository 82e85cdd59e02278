// (1) Weight-threshold fine-sample selection + compaction, entirely on the device:
//     replaces nonzero / expand / arithmetic / index_put of model/mc_nerf.py:623-629, 692-694 and
//     removes the reference's host syncs (.item(), nonzero).
// (2) Per-pixel ray generation from one camera's world->cam pose and inverse intrinsics, forward and
//     backward: replaces MC_Model.get_rays + generate_rand_rays (model/mc_nerf.py:124-145, 213-256,
//     327-345) for the selected pixels only.
#include "mcnerf_kernels.h"

// ------------------------------------------------------------------ selection
__device__ __forceinline__ float sel_threshold(const McnSelectArgs& a) {
    // min(weight_thresh, weights.max()) (model/mc_nerf.py:623)
    return fminf(a.thresh, __uint_as_float(*a.wmax_bits));
}

__global__ __launch_bounds__(256) void select_count_kernel(McnSelectArgs a) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= a.N) return;
    const float thr = sel_threshold(a);
    int cnt = 0;
    for (int base = 0; base < a.Sc; base += 64) {
        const int j = base + lane;
        const bool sel = j < a.Sc && a.w_sel[(size_t)n * a.Sc + j] >= thr;
        cnt += __popcll(__ballot(sel));
    }
    if (lane == 0) a.ray_counts[n] = cnt * a.scale;
    if (a.out_f) {      // defaults for never-evaluated fine samples (model/mc_nerf.py:692-694)
        const int Sf = a.Sc * a.scale;
        f32x4 d; d[0] = a.sigma_default; d[1] = 1.f; d[2] = 1.f; d[3] = 1.f;
        f32x4* o = reinterpret_cast<f32x4*>(a.out_f) + (size_t)n * Sf;
        for (int j = lane; j < Sf; j += 64) o[j] = d;
    }
}

// Exclusive scan of ray_counts -> ray_offsets, total -> *count.  One workgroup; N is at most ~1e6.  Every thread owns one
// contiguous run of `per` rays per thread: a private sum, ONE block-wide scan of the 1024 run sums, then the run's offsets -- two
// barriers in all.  A run is a multiple of four rays held in registers as int4 vectors: its loads (and its stores) are independent
// 16-byte operations in flight together (the scalar loop of dependent 4-byte loads over a 32-ray run took 53 us at 32768 rays);
// runs longer than 64 rays (N > 65536) or unaligned workspaces take the scalar loop.
__global__ __launch_bounds__(1024) void select_scan_kernel(McnSelectArgs a) {
    __shared__ int wsum[16];
    constexpr int MAXQ = 16;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int per = (((a.N + 1023) / 1024) + 3) & ~3;
    const int lo = min(tid * per, a.N), hi = min(lo + per, a.N);
    const bool vec = per <= 4 * MAXQ && ((reinterpret_cast<size_t>(a.ray_counts) | reinterpret_cast<size_t>(a.ray_offsets)) & 15) == 0;
    int4 v[MAXQ];
    int s = 0;
    if (vec) {
#pragma unroll
        for (int q = 0; q < MAXQ; ++q) {
            const int b = lo + 4 * q;
            int4 x = make_int4(0, 0, 0, 0);
            if (4 * q < per) {
                if (b + 3 < hi) x = *reinterpret_cast<const int4*>(a.ray_counts + b);
                else {
                    if (b < hi) x.x = a.ray_counts[b];
                    if (b + 1 < hi) x.y = a.ray_counts[b + 1];
                    if (b + 2 < hi) x.z = a.ray_counts[b + 2];
                }
            }
            v[q] = x;
        }
#pragma unroll
        for (int q = 0; q < MAXQ; ++q) s += (v[q].x + v[q].y) + (v[q].z + v[q].w);
    } else {
        for (int i = lo; i < hi; ++i) s += a.ray_counts[i];
    }
    int inc = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    int woff = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { const int t = wsum[w]; woff += w < wv ? t : 0; total += t; }
    int run = woff + inc - s;
    if (vec) {
#pragma unroll
        for (int q = 0; q < MAXQ; ++q) {
            const int b = lo + 4 * q;
            if (4 * q < per && b < hi) {
                int4 o;
                o.x = run; run += v[q].x; o.y = run; run += v[q].y; o.z = run; run += v[q].z; o.w = run; run += v[q].w;
                if (b + 3 < hi) *reinterpret_cast<int4*>(a.ray_offsets + b) = o;
                else {
                    a.ray_offsets[b] = o.x;
                    if (b + 1 < hi) a.ray_offsets[b + 1] = o.y;
                    if (b + 2 < hi) a.ray_offsets[b + 2] = o.z;
                }
            }
        }
    } else {
        for (int i = lo; i < hi; ++i) { const int t = a.ray_counts[i]; a.ray_offsets[i] = run; run += t; }
    }
    if (tid == 0) *a.count = total;
}

__global__ __launch_bounds__(256) void select_write_kernel(McnSelectArgs a) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= a.N) return;
    const float thr = sel_threshold(a);
    int pos = a.ray_offsets[n];
    for (int base = 0; base < a.Sc; base += 64) {
        const int j = base + lane;
        const bool sel = j < a.Sc && a.w_sel[(size_t)n * a.Sc + j] >= thr;
        const unsigned long long m = __ballot(sel);
        if (sel) {
            const int rank = __popcll(m & ((1ull << lane) - 1ull));
            int2* dst = a.idx + pos + rank * a.scale;
            for (int r = 0; r < a.scale; ++r) dst[r] = make_int2(n, j * a.scale + r);
        }
        pos += __popcll(m) * a.scale;
    }
}

hipError_t mcn_launch_select(const McnSelectArgs& a, hipStream_t st) {
    if (a.N <= 0) return hipSuccess;
    const dim3 g((a.N + 3) / 4), b(256);
    hipLaunchKernelGGL(select_count_kernel, g, b, 0, st, a);
    hipLaunchKernelGGL(select_scan_kernel, dim3(1), dim3(1024), 0, st, a);
    hipLaunchKernelGGL(select_write_kernel, g, b, 0, st, a);
    return hipGetLastError();
}

// Keeps idx_in[perm[i]], i < keep: the random cap of model/mc_nerf.py:630-632.
__global__ void cap_gather_kernel(const int2* idx_in, const long long* perm, int keep, int2* idx_out, int* count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < keep) idx_out[i] = idx_in[perm[i]];
    if (i == 0) *count = keep;
}
hipError_t mcn_launch_cap_gather(const int2* idx_in, const long long* perm, int keep, int2* idx_out, int* count, hipStream_t st) {
    if (keep <= 0) return hipSuccess;
    hipLaunchKernelGGL(cap_gather_kernel, dim3((keep + 255) / 256), dim3(256), 0, st, idx_in, perm, keep, idx_out, count);
    return hipGetLastError();
}

// ---- The random cap of model/mc_nerf.py:630-632 without a host round trip.  The reference keeps idx[randperm(K)[:keep]]
// when K > keep: a uniformly random subset of size `keep` (the order of the list never reaches a result).  Here every
// entry i < K gets a 32-bit key hash(seed, i) and the `keep` smallest keys are kept: two 65536-bin histogram passes find
// the exact threshold key, a counting pass and an ordered compaction write the kept entries in list order.  K <= keep keeps everything.
// ws (uint32): [0 .. 65535] histogram of the high 16 key bits, [65536 .. 131071] histogram of the low 16 bits inside the
// boundary bin, [131072] boundary bin (0x10000 = keep all), [131073] entries below it, [131074] threshold key,
// [131075] ties to take at the threshold, [MCN_CAP_LT + b] / [MCN_CAP_EQ + b] entries of chunk b below / at the threshold.
__device__ __forceinline__ unsigned cap_key(unsigned seed, unsigned i) {
    unsigned x = i * 0x9E3779B9u + seed;          // murmur3 finaliser: every output bit depends on every input bit
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}
__global__ __launch_bounds__(256) void cap_hist_hi_kernel(const int* count, int max_rows, int keep, const unsigned* seed, unsigned* ws) {
    const int K = min(*count, max_rows);
    if (K <= keep) return;
    const unsigned sd = *seed;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < K; i += gridDim.x * blockDim.x) atomicAdd(&ws[cap_key(sd, i) >> 16], 1u);
}
// one workgroup: first bin whose inclusive prefix reaches `target`; writes (bin, prefix before it)
__device__ void cap_find(const unsigned* hist, unsigned target, unsigned* out_bin, unsigned* out_below) {
    __shared__ unsigned part[1024];
    const int tid = threadIdx.x;
    unsigned s = 0;
    for (int b = 0; b < 64; ++b) s += hist[tid * 64 + b];
    part[tid] = s;
    __syncthreads();
    if (tid == 0) {
        unsigned cum = 0;
        int t = 0;
        while (t < 1024 && cum + part[t] < target) { cum += part[t]; ++t; }
        int b = t * 64;
        if (t < 1024) { while (cum + hist[b] < target) { cum += hist[b]; ++b; } }
        *out_bin = (unsigned)b; *out_below = cum;
    }
    __syncthreads();
}
__global__ __launch_bounds__(1024) void cap_find_hi_kernel(const int* count, int max_rows, int keep, unsigned* ws) {
    const int K = min(*count, max_rows);
    if (K <= keep) { if (threadIdx.x == 0) { ws[131072] = 0x10000u; ws[131074] = 0xFFFFFFFFu; ws[131075] = 0xFFFFFFFFu; } return; }
    cap_find(ws, (unsigned)keep, &ws[131072], &ws[131073]);
}
__global__ __launch_bounds__(256) void cap_hist_lo_kernel(const int* count, int max_rows, int keep, const unsigned* seed, unsigned* ws) {
    const int K = min(*count, max_rows);
    if (K <= keep) return;
    const unsigned sd = *seed, bin = ws[131072];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < K; i += gridDim.x * blockDim.x) {
        const unsigned k = cap_key(sd, i);
        if ((k >> 16) == bin) atomicAdd(&ws[65536 + (k & 0xFFFFu)], 1u);
    }
}
__global__ __launch_bounds__(1024) void cap_find_lo_kernel(const int* count, int max_rows, int keep, unsigned* ws) {
    const int K = min(*count, max_rows);
    if (K <= keep) return;
    __shared__ unsigned lo, below;
    cap_find(ws + 65536, (unsigned)keep - ws[131073], &lo, &below);
    if (threadIdx.x == 0) {
        ws[131074] = (ws[131072] << 16) | lo;                        // threshold key: all smaller keys are kept
        ws[131075] = (unsigned)keep - ws[131073] - below;            // ... and this many entries with exactly that key
    }
}
// The kept entries keep their order in the list (the reference's order whenever the cap does not bind, and a list that is the same,
// entry for entry, from run to run when it does: no atomic decides a position): every workgroup owns one contiguous chunk of
// the list, counts what it keeps (keys below the threshold, keys equal to it), and writes behind the chunks before it; of the entries
// whose key EQUALS the threshold the first `ties` in list order are taken.
__device__ __forceinline__ int cap_chunk(int K, int grid) { return (((K + grid - 1) / grid) + 255) & ~255; }
__global__ __launch_bounds__(256) void cap_count_kernel(const int* count, int max_rows, int keep, const unsigned* seed, unsigned* ws) {
    const int K = min(*count, max_rows);
    if (K <= keep) return;
    const unsigned sd = *seed, thr = ws[131074];
    const int chunk = cap_chunk(K, gridDim.x), lo = blockIdx.x * chunk, hi = min(K, lo + chunk);
    unsigned lt = 0, eq = 0;
    for (int i = lo + threadIdx.x; i < hi; i += 256) {
        const unsigned k = cap_key(sd, (unsigned)i);
        lt += k < thr; eq += k == thr;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { lt += __shfl_xor(lt, o); eq += __shfl_xor(eq, o); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&ws[MCN_CAP_LT + blockIdx.x], lt); atomicAdd(&ws[MCN_CAP_EQ + blockIdx.x], eq); }     // (integer sums: order-free)
}
// exclusive rank of `flag` among the 256 threads of the workgroup (in thread order) and the workgroup's total
__device__ __forceinline__ unsigned cap_block_rank(bool flag, unsigned* wsum, unsigned& total) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned long long m = __ballot(flag);
    __syncthreads();
    if (lane == 0) wsum[wv] = (unsigned)__popcll(m);
    __syncthreads();
    unsigned before = 0;
    total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { const unsigned v = wsum[w]; before += w < wv ? v : 0u; total += v; }
    return before + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
}
__global__ __launch_bounds__(256) void cap_write_kernel(const int2* idx_in, const int* count, int max_rows, int keep, const unsigned* seed,
                                                        unsigned* ws, int2* idx_out, int* count_out) {
    __shared__ unsigned wsum[4], red[8];
    const int K = min(*count, max_rows);
    if (blockIdx.x == 0 && threadIdx.x == 0) *count_out = min(K, keep);
    const int chunk = cap_chunk(K, gridDim.x), lo = blockIdx.x * chunk, hi = min(K, lo + chunk);
    if (K <= keep) {                                  // nothing to drop: the list as it stands
        for (int i = lo + threadIdx.x; i < hi; i += 256) idx_out[i] = idx_in[i];
        return;
    }
    const unsigned sd = *seed, thr = ws[131074], ties = ws[131075];
    unsigned lt = 0, eq = 0;                          // kept / tied entries of the chunks in front of this one
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += 256) { lt += ws[MCN_CAP_LT + b]; eq += ws[MCN_CAP_EQ + b]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { lt += __shfl_xor(lt, o); eq += __shfl_xor(eq, o); }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = lt; red[4 + (threadIdx.x >> 6)] = eq; }
    __syncthreads();
    lt = red[0] + red[1] + red[2] + red[3];
    unsigned eq_run = red[4] + red[5] + red[6] + red[7];
    unsigned pos = lt + min(eq_run, ties);
    for (int base = lo; base < hi; base += 256) {
        const int i = base + threadIdx.x;
        const unsigned k = i < hi ? cap_key(sd, (unsigned)i) : 0xFFFFFFFFu;
        const bool is_eq = i < hi && k == thr;
        unsigned n_eq, n_sel;
        const unsigned tie_rank = cap_block_rank(is_eq, wsum, n_eq);
        const bool sel = i < hi && (k < thr || (is_eq && eq_run + tie_rank < ties));
        const unsigned rank = cap_block_rank(sel, wsum, n_sel);
        if (sel) idx_out[pos + rank] = idx_in[i];
        pos += n_sel; eq_run += n_eq;
    }
}
hipError_t mcn_launch_cap_random(const int2* idx_in, const int* count, int max_rows, int keep, const unsigned* seed, unsigned* ws,
                                 int2* idx_out, int* count_out, hipStream_t st) {
    if (max_rows <= 0 || keep <= 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(ws, 0, MCN_CAP_WS * sizeof(unsigned), st);
    if (e != hipSuccess) return e;
    const int blocks = min((max_rows + 255) / 256, MCN_CAP_BLOCKS);
    hipLaunchKernelGGL(cap_hist_hi_kernel, dim3(blocks), dim3(256), 0, st, count, max_rows, keep, seed, ws);
    hipLaunchKernelGGL(cap_find_hi_kernel, dim3(1), dim3(1024), 0, st, count, max_rows, keep, ws);
    hipLaunchKernelGGL(cap_hist_lo_kernel, dim3(blocks), dim3(256), 0, st, count, max_rows, keep, seed, ws);
    hipLaunchKernelGGL(cap_find_lo_kernel, dim3(1), dim3(1024), 0, st, count, max_rows, keep, ws);
    hipLaunchKernelGGL(cap_count_kernel, dim3(blocks), dim3(256), 0, st, count, max_rows, keep, seed, ws);
    hipLaunchKernelGGL(cap_write_kernel, dim3(blocks), dim3(256), 0, st, idx_in, count, max_rows, keep, seed, ws, idx_out, count_out);
    return hipGetLastError();
}

// ------------------------------------------------------------------ stand-alone positional encoding
// SinCosEmbedding.forward (model/net_block.py:20-35): [n,3] -> [n,3+6F] = [x, per axis sin(2^f x) f<F, cos(2^f x) f<F] (F = 10: 63)
// times the per-frequency BARF weights.  (The render path computes the same inside the fused MLP kernels.)
// (F = `emb_freqs_xyz` frequencies, 3 + 6 F output channels; one thread per (point, axis, frequency), F = 0: per (point, axis))
__global__ __launch_bounds__(256) void encode_kernel(const float* x, const float* barf_w, int n, int F, float* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int per = 3 * (F > 0 ? F : 1), nenc = 3 + 6 * F;
    if (i >= n * per) return;
    const int m = i / per, cf = i - m * per, c = F > 0 ? cf / F : cf, f = F > 0 ? cf - c * F : 0;
    const float xv = x[m * 3 + c];
    if (f == 0) out[(size_t)m * nenc + c] = xv;
    if (F == 0) return;
    float s, co;
    mcn_sincos(xv * (float)(1 << f), s, co);
    const float w = barf_w[f];
    out[(size_t)m * nenc + 3 + c * 2 * F + f] = s * w;
    out[(size_t)m * nenc + 3 + c * 2 * F + F + f] = co * w;
}
hipError_t mcn_launch_encode(const float* x, const float* barf_w, int n, int n_freqs, float* out, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    const int per = 3 * (n_freqs > 0 ? n_freqs : 1);
    hipLaunchKernelGGL(encode_kernel, dim3((n * per + 255) / 256), dim3(256), 0, st, x, barf_w, n, n_freqs, out);
    return hipGetLastError();
}
// its backward: d x_c = d out_c + sum_f w_f 2^f (cos(2^f x_c) d sin_f - sin(2^f x_c) d cos_f); one thread per (point, axis)
__global__ __launch_bounds__(256) void encode_bwd_kernel(const float* x, const float* barf_w, int n, int F, const float* d_out, float* d_x) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * 3) return;
    const int m = i / 3, c = i - m * 3;
    const float xv = x[i];
    const float* g = d_out + (size_t)m * (3 + 6 * F);
    float dx = g[c];
    for (int f = 0; f < F; ++f) {
        float s, co;
        mcn_sincos(xv * (float)(1 << f), s, co);
        dx = fmaf(barf_w[f] * (float)(1 << f), co * g[3 + c * 2 * F + f] - s * g[3 + c * 2 * F + F + f], dx);
    }
    d_x[i] = dx;
}
hipError_t mcn_launch_encode_bwd(const float* x, const float* barf_w, int n, int n_freqs, const float* d_out, float* d_x, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(encode_bwd_kernel, dim3((n * 3 + 255) / 256), dim3(256), 0, st, x, barf_w, n, n_freqs, d_out, d_x);
    return hipGetLastError();
}

// ---- The pixel subset of a train step: randperm(H * W)[:batch] (model/mc_nerf.py:329, a uniformly random ordered subset
// without replacement) as `batch` evaluations of a keyed pseudo-random PERMUTATION of [0, n): a 6-round balanced Feistel
// network on 2 * half bits (the smallest even width covering n) with the murmur finaliser as round function, cycle-walked
// back into [0, n) (the domain is < 4 n, so < 4 walks on average).  One 7 us kernel instead of the 22 kernels of a
// device randperm of 640 000 keys (radix sort + merges, 0.2 ms/step).
__global__ __launch_bounds__(256) void sample_perm_kernel(long long* out, unsigned n, int batch, const unsigned* seed) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= batch) return;
    int bits = 1;
    while (bits < 32 && (1ull << bits) < n) ++bits;
    const int half = (bits + 1) / 2;
    const unsigned mask = (1u << half) - 1u, sd = *seed;
    unsigned x = (unsigned)i;
    do {
        unsigned L = x >> half, R = x & mask;
#pragma unroll
        for (unsigned r = 0; r < 6; ++r) {
            const unsigned f = cap_key(sd + 0x632BE5ABu * (r + 1), R) & mask;
            const unsigned nl = R;
            R = L ^ f; L = nl;
        }
        x = (L << half) | R;
    } while (x >= n);
    out[i] = (long long)x;
}
hipError_t mcn_launch_sample_perm(long long* out, long long n, int batch, const unsigned* seed, hipStream_t st) {
    if (batch <= 0) return hipSuccess;
    hipLaunchKernelGGL(sample_perm_kernel, dim3((batch + 255) / 256), dim3(256), 0, st, out, (unsigned)n, batch, seed);
    return hipGetLastError();
}

// up to 16 host floats by value (kernel arguments) -> device memory: a stream-ordered upload that never blocks the host
struct McnFloats16 { float v[16]; };
__global__ void upload_f32_kernel(float* dst, McnFloats16 vals, int n) {
    if ((int)threadIdx.x < n) dst[threadIdx.x] = vals.v[threadIdx.x];
}
hipError_t mcn_launch_upload_f32(float* dst, const float* host_vals, int n, hipStream_t st) {
    McnFloats16 v = {};
    for (int i = 0; i < n && i < 16; ++i) v.v[i] = host_vals[i];
    hipLaunchKernelGGL(upload_f32_kernel, dim3(1), dim3(64), 0, st, dst, v, n);
    return hipGetLastError();
}

// ------------------------------------------------------------------ ray generation
// d = normalize(R^T K^-1 [u+.5, v+.5, 1]^T), o = -R^T t, following the reference's op order
// (pix @ K^-T, lift, @ pose_inv^T, minus origin, normalise) so results agree to ~1e-7.
__global__ __launch_bounds__(256) void raygen_fwd_kernel(McnRaygenArgs a) {
    __shared__ float P[12], K[9];
    if (threadIdx.x < 12) P[threadIdx.x] = a.pose[threadIdx.x];
    if (threadIdx.x < 9) K[threadIdx.x] = a.kinv[threadIdx.x];
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const long long pid = a.pix[i];
    const float u = (float)(pid % a.W) + 0.5f, v = (float)(pid / a.W) + 0.5f;
    float cam[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) cam[r] = __fadd_rn(__fadd_rn(__fmul_rn(u, K[r * 3]), __fmul_rn(v, K[r * 3 + 1])), K[r * 3 + 2]);
    float d[3], o[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        // pose_inv row c = [R[0][c], R[1][c], R[2][c], -(R^T t)[c]]
        const float ti = -(__fadd_rn(__fadd_rn(__fmul_rn(P[0 * 4 + c], P[3]), __fmul_rn(P[1 * 4 + c], P[7])), __fmul_rn(P[2 * 4 + c], P[11])));
        const float w = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(cam[0], P[0 * 4 + c]), __fmul_rn(cam[1], P[1 * 4 + c])), __fmul_rn(cam[2], P[2 * 4 + c])), ti);
        o[c] = ti;
        d[c] = __fsub_rn(w, ti);
    }
    const float nrm = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
#pragma unroll
    for (int c = 0; c < 3; ++c) { a.rays_d[i * 3 + c] = d[c] / nrm; a.rays_o[i * 3 + c] = o[c]; }
}

hipError_t mcn_launch_raygen_fwd(const McnRaygenArgs& a, hipStream_t st) {
    if (a.n <= 0) return hipSuccess;
    hipLaunchKernelGGL(raygen_fwd_kernel, dim3((a.n + 255) / 256), dim3(256), 0, st, a);
    return hipGetLastError();
}

// Backward: accumulates d_pose[3][4] and d_kinv[3][3] over the n rays (block reduction + one atomic
// per block and value).
__global__ __launch_bounds__(256) void raygen_bwd_kernel(McnRaygenBwdArgs a) {
    __shared__ float P[12], K[9];
    __shared__ float red[4][24];
    if (threadIdx.x < 12) P[threadIdx.x] = a.pose[threadIdx.x];
    if (threadIdx.x < 9) K[threadIdx.x] = a.kinv[threadIdx.x];
    __syncthreads();
    float acc[21];
#pragma unroll
    for (int k = 0; k < 21; ++k) acc[k] = 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += gridDim.x * blockDim.x) {
        const long long pid = a.pix[i];
        const float p[3] = {(float)(pid % a.W) + 0.5f, (float)(pid / a.W) + 0.5f, 1.f};
        float cam[3], q[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) cam[r] = p[0] * K[r * 3] + p[1] * K[r * 3 + 1] + K[r * 3 + 2];
#pragma unroll
        for (int c = 0; c < 3; ++c) q[c] = cam[0] * P[c] + cam[1] * P[4 + c] + cam[2] * P[8 + c];
        const float inv = 1.f / sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
        const float gd[3] = {a.d_rays_d[i * 3], a.d_rays_d[i * 3 + 1], a.d_rays_d[i * 3 + 2]};
        const float dn[3] = {q[0] * inv, q[1] * inv, q[2] * inv};
        const float dot = dn[0] * gd[0] + dn[1] * gd[1] + dn[2] * gd[2];
        float gq[3], gcam[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) gq[c] = (gd[c] - dn[c] * dot) * inv;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            gcam[j] = P[j * 4] * gq[0] + P[j * 4 + 1] * gq[1] + P[j * 4 + 2] * gq[2];
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[j * 3 + c] += cam[j] * gq[c];        // dR[j][c] from the direction
#pragma unroll
            for (int k = 0; k < 3; ++k) acc[9 + j * 3 + k] += gcam[j] * p[k];     // dKinv[j][k]
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[18 + c] += a.d_rays_o[i * 3 + c];         // sum of origin gradients
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 21; ++k) {
        float v = acc[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) red[wv][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 21) {
        const int k = threadIdx.x;
        const float v = red[0][k] + red[1][k] + red[2][k] + red[3][k];
        if (k < 9) atomicAdd(&a.d_pose[(k / 3) * 4 + (k % 3)], v);
        else if (k < 18) atomicAdd(&a.d_kinv[k - 9], v);
        else {
            // o_c = -sum_j R[j][c] t_j:  dR[j][c] += -t_j * Go_c ;  dt_j = -sum_c R[j][c] Go_c
            const int c = k - 18;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                atomicAdd(&a.d_pose[j * 4 + c], -P[j * 4 + 3] * v);
                atomicAdd(&a.d_pose[j * 4 + 3], -P[j * 4 + c] * v);
            }
        }
    }
}

hipError_t mcn_launch_raygen_bwd(const McnRaygenBwdArgs& a, hipStream_t st) {
    if (a.n <= 0) return hipSuccess;
    int grid = (a.n + 255) / 256;
    if (grid > 512) grid = 512;
    hipLaunchKernelGGL(raygen_bwd_kernel, dim3(grid), dim3(256), 0, st, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------ device-resident images
// GT colour of the selected pixels of one camera straight from uint8 images kept in HBM:
//   rgb = rgb8/255 * a + (1 - a),  a = alpha8/255   (RGBA composited on white, data/data_read.py:130-137;
//   ToTensor's /255 first, then the blend in fp32, as the reference does), or rgb8/255 for 3-channel images.
// Replaces the 7.7 MB/step H2D copy of a float image plus `gt_rgbs.reshape(-1,3)[rand_idx]` (model/mc_nerf.py:379, 80).
__global__ __launch_bounds__(256) void gather_gt_kernel(const unsigned char* __restrict__ img, int channels,
                                                        const long long* __restrict__ pix, int n, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned char* p = img + (size_t)pix[i] * channels;
    const float r = (float)p[0] / 255.0f, g = (float)p[1] / 255.0f, b = (float)p[2] / 255.0f;
    if (channels == 4) {
        const float a = (float)p[3] / 255.0f;
        out[i * 3 + 0] = r * a + (1.0f - a);
        out[i * 3 + 1] = g * a + (1.0f - a);
        out[i * 3 + 2] = b * a + (1.0f - a);
    } else {
        out[i * 3 + 0] = r; out[i * 3 + 1] = g; out[i * 3 + 2] = b;
    }
}
hipError_t mcn_launch_gather_gt(const unsigned char* img, int channels, const long long* pix, int n, float* out, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(gather_gt_kernel, dim3((n + 255) / 256), dim3(256), 0, st, img, channels, pix, n, out);
    return hipGetLastError();
}
