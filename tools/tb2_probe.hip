// Feasibility probe: the memory access pattern of a two-steps-per-pass 2-D sweep
// (wave = 64 float4 words wide, 60 of them owned, marching R + 2 rows for R owned rows;
// 6 arrays read, 6 written per pass) with placeholder arithmetic.  Prints ms per pass
// and the equivalent cell-updates/s for a 16384 x 16384 grid.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));

struct Arrays {
    const f4 *hx, *hy, *ez0, *ezm, *dz0, *dzm;   // read
    f4 *hxn, *hyn, *ez1, *ez2, *dz1, *dz2;       // written (separate arrays: halo rows are re-read by neighbours)
    int words;      // float4 words per row
    int rows;
    int strips;     // waves across a row
};

template <bool NT> __device__ inline f4 ld(const f4 *p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ inline void st(f4 v, f4 *p) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

__device__ inline f4 shl(f4 v) { f4 r; r.x = v.y; r.y = v.z; r.z = v.w; r.w = __shfl_down(v.x, 1); return r; }
__device__ inline f4 shr(f4 v) { f4 r; r.y = v.x; r.z = v.y; r.w = v.z; r.x = __shfl_up(v.w, 1); return r; }

// ALT: odd bands march downwards (so both neighbours touch the rows they share at the same time);
// PAD: blocks per row of bands padded to a multiple of 8 (vertical neighbours on one XCD).
template <int R, bool NT, int K, bool ALT, bool PAD, int OWN, int LEFT> __global__ __launch_bounds__(256) void tb2(Arrays a)
{
    const int lane = threadIdx.x & 63;
    const int bpr = (a.strips + 3) / 4, bprp = PAD ? (bpr + 7) / 8 * 8 : bpr;
    const int xb = blockIdx.x % bprp;
    const long band = blockIdx.x / bprp;
    if (xb >= bpr) return;
    const int strip = xb * 4 + (threadIdx.x >> 6);
    if (strip >= a.strips) return;
    const long y0 = band * R;
    if (y0 >= a.rows) return;
    const bool down = ALT && (band & 1);
    int w = strip * OWN - LEFT + lane;
    const bool inrow = w >= 0 && w < a.words;
    const bool own = lane >= LEFT && lane < LEFT + OWN && inrow;
    if (!inrow) w = w < 0 ? 0 : a.words - 1;
    // carried state of the previous row
    f4 hx1p = {0, 0, 0, 0}, hy1p = hx1p, ez1p = hx1p, ez0p = hx1p, dz1p = hx1p, dz0p = hx1p, hx2pp = hx1p;
    long y = y0 - K;
    if (y < 0) y = 0;
    f4 ez0 = ld<false>(a.ez0 + y * a.words + w);
    for (long jj = y0 - K; jj <= y0 + R + K - 2; ++jj) {
        const long j = down ? (y0 - K) + (y0 + R + K - 2) - jj : jj;
        if (j >= a.rows) continue;
        const long row = j < 0 ? 0 : j;
        const size_t o = (size_t)row * a.words + w;
        const long up = row + 1 < a.rows ? row + 1 : row;
        const f4 ez0u = ld<false>(a.ez0 + (size_t)up * a.words + w);
        const f4 hx = ld<NT>(a.hx + o), hy = ld<NT>(a.hy + o), ezm = ld<NT>(a.ezm + o), dz0 = ld<NT>(a.dz0 + o),
                 dzm = ld<NT>(a.dzm + o);
        // step 1 on this row
        const f4 hx1 = hx - 0.5f * (ez0u - ez0);
        const f4 hy1 = hy + 0.5f * (shl(ez0) - ez0);
        const f4 dz1 = dz0 + 0.5f * (hy1 - shr(hy1)) - 0.5f * (hx1 - hx1p);
        const f4 ez1 = 0.3f * ez0 - 0.2f * ezm + 0.9f * dz1 - 0.4f * dz0 + 0.1f * dzm;
        // step 2 on the previous row
        const f4 hx2 = hx1p - 0.5f * (ez1 - ez1p);
        const f4 hy2 = hy1p + 0.5f * (shl(ez1p) - ez1p);
        const f4 dz2 = dz1p + 0.5f * (hy2 - shr(hy2)) - 0.5f * (hx2 - hx2pp);
        const f4 ez2 = 0.3f * ez1p - 0.2f * ez0p + 0.9f * dz2 - 0.4f * dz1p + 0.1f * dz0p;
        const long out = down ? j + 1 : j - 1;
        if (own && out >= y0 && out < y0 + R && out < a.rows) {
            const size_t q = (size_t)out * a.words + w;
            st<NT>(hx2, a.hxn + q);
            st<NT>(hy2, a.hyn + q);
            st<NT>(ez1p, a.ez1 + q);
            st<NT>(ez2, a.ez2 + q);
            st<NT>(dz1p, a.dz1 + q);
            st<NT>(dz2, a.dz2 + q);
        }
        hx2pp = hx2; hx1p = hx1; hy1p = hy1; ez1p = ez1; ez0p = ez0; dz1p = dz1; dz0p = dz0;
        ez0 = ez0u;
    }
}

template <int R, bool NT, int K, bool ALT, bool PAD, int OWN, int LEFT> void run(Arrays a)
{
    a.strips = (a.words + OWN - 1) / OWN;
    const long bands = (a.rows + R - 1) / R;
    const int bpr = (a.strips + 3) / 4, bprp = PAD ? (bpr + 7) / 8 * 8 : bpr;
    const unsigned blocks = (unsigned)(bands * bprp);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((tb2<R, NT, K, ALT, PAD, OWN, LEFT>), dim3(blocks), dim3(256), 0, 0, a);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((tb2<R, NT, K, ALT, PAD, OWN, LEFT>), dim3(blocks), dim3(256), 0, 0, a);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    printf("R=%2d nt=%d K=%d own=%d left=%d: %.3f ms per pass\n", R, (int)NT, K, OWN, LEFT, ms);
    fflush(stdout);
}

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 16384;      // columns
    const int nrows = argc > 2 ? atoi(argv[2]) : n;
    const size_t cells = (size_t)n * nrows;
    // one arena for all arrays: array k starts k * (array bytes rounded up to `quantum`, plus `pad`) into it
    const size_t pad = argc > 3 ? (size_t)atol(argv[3]) : 4096;
    const size_t quantum = argc > 4 ? (size_t)atol(argv[4]) : 256;
    const size_t stride = (cells * 4 + quantum - 1) / quantum * quantum + pad;
    char *arena;
    hipMalloc((void **)&arena, 12 * stride + 4096);
    hipMemset(arena, 0, 12 * stride + 4096);
    float *buf[12];
    for (int k = 0; k < 12; ++k) buf[k] = (float *)(arena + k * stride);
    printf("stride %zu bytes (pad %zu)\n", stride, pad);
    Arrays a;
    a.hx = (f4 *)buf[0]; a.hxn = (f4 *)buf[1];
    a.hy = (f4 *)buf[2]; a.hyn = (f4 *)buf[3];
    a.ez0 = (f4 *)buf[4]; a.ezm = (f4 *)buf[5]; a.ez1 = (f4 *)buf[6]; a.ez2 = (f4 *)buf[7];
    a.dz0 = (f4 *)buf[8]; a.dzm = (f4 *)buf[9]; a.dz1 = (f4 *)buf[10]; a.dz2 = (f4 *)buf[11];
    a.words = n / 4;
    a.rows = nrows;
    a.strips = (a.words + 61) / 62;
    run<64, true, 8, false, false, 60, 2>(a);
    run<32, true, 4, false, false, 60, 2>(a);
    return 0;
}
