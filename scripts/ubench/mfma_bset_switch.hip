// Microbenchmark (r02): does the FIRST tile that takes its MFMA B operands from a different register set than the
// previous tiles run slower, and does it depend on where the sets live (ArchVGPR / AccVGPR)?
// Signature being chased: in the resident light-visibility kernel and in the NeRF kernel the first tile of every layer
// costs 2-3x a steady tile; profiles/r02/lv2_x.log shows it follows the SWITCH of the B register set (a layer that
// keeps reading the previous layer's B registers has a fast first tile) and survives a 30 k-cycle sleep.
// One wave per SIMD, 4 column tiles, 4 k-steps per tile (16 MFMAs = 512 cycles), A operands register-resident.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_bset_switch.hip -o mfma_bset_switch.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MFMA_V(ACC, A, B) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "v"(B))
#define MFMA_A(ACC, A, B) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "a"(B))

constexpr int NT = 16;   // tiles: X X X X Y Y Y Y X X X X Y Y Y Y

// MODE 5 / 6: as 0, but Y receives NEW VALUES (converted from the accumulators) before some of its tiles.
// MODE 0: X, Y in ArchVGPRs; 1: X Arch, Y Acc; 2: X, Y in AccVGPRs; 3: as 2, Y rewritten (v_accvgpr_write) before its
// first tile of the second round; 4: as 0 with a single set (no switch) as the control
template <int MODE>
__global__ __launch_bounds__(256, 1) void k(const float* in, float* out, unsigned long long* t) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[4];
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    bf16x8 a[4], bx[4][4], by[4][4];
    for (int s = 0; s < 4; ++s) {
        for (int j = 0; j < 8; ++j) a[s][j] = (__bf16)in[(lane + s * 8 + j) & 1023];
        for (int c = 0; c < 4; ++c)
            for (int j = 0; j < 8; ++j) {
                bx[s][c][j] = (__bf16)in[(lane * 3 + s + c * 7 + j) & 1023];
                by[s][c][j] = (__bf16)in[(lane * 5 + s * 3 + c + j) & 1023];
            }
    }
    unsigned long long ts[NT + 1];
#pragma unroll
    for (int tile = 0; tile < NT; ++tile) {
        const bool useY = MODE != 4 && ((tile >> 2) & 1);
        if (MODE == 3 && tile == 12) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
                    u32x4 v = __builtin_bit_cast(u32x4, by[s][c]);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {   // VALU touch of every dword, then back into the Acc file
                        unsigned int w = v[q];
                        asm volatile("v_pk_max_i16 %0, %0, 0" : "+v"(w));
                        v[q] = w;
                    }
                    by[s][c] = __builtin_bit_cast(bf16x8, v);
                }
        }
        if ((MODE == 5 && tile == 12) || (MODE == 6 && (tile == 5 || tile == 6 || tile == 12 || tile == 13))) {
            // NEW VALUES in Y (not the same bits written again): converted from floats, as an epilogue would
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        by[s][c][j] = (__bf16)(acc[c][(j + s + tile) & 15] * 1e-3f + (float)by[s][c][j] * 0.5f);
        }
        __builtin_amdgcn_sched_barrier(0);
        ts[tile] = __builtin_readcyclecounter();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (!useY) {
                    if (MODE >= 2 && MODE < 4) MFMA_A(acc[c], a[s], bx[s][c]);
                    else MFMA_V(acc[c], a[s], bx[s][c]);
                } else {
                    if (MODE >= 1 && MODE < 5) MFMA_A(acc[c], a[s], by[s][c]);
                    else MFMA_V(acc[c], a[s], by[s][c]);
                }
            }
        __builtin_amdgcn_sched_barrier(0);
    }
    ts[NT] = __builtin_readcyclecounter();
    float sum = 0.f;
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 16; ++r) sum += acc[c][r];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
    if (blockIdx.x == 3 && threadIdx.x == 0)
        for (int i = 0; i <= NT; ++i) t[i] = ts[i];
}

template <int MODE>
static void run(const float* in, float* out, unsigned long long* t, const char* what) {
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, in, out, t);
    hipDeviceSynchronize();
    unsigned long long ht[NT + 1];
    hipMemcpy(ht, t, sizeof(ht), hipMemcpyDeviceToHost);
    printf("mode %d (%s): cycles per 16-MFMA tile (ideal 512):", MODE, what);
    for (int i = 0; i < NT; ++i) printf("%s%llu", (i % 4) ? " " : " | ", ht[i + 1] - ht[i]);
    printf("\n");
}

int main() {
    float *in, *out;
    unsigned long long* t;
    hipMalloc(&in, 4096);
    hipMalloc(&out, 256 * 256 * 4);
    hipMalloc(&t, 256);
    float h[1024];
    for (int i = 0; i < 1024; ++i) h[i] = (i % 37) * 0.01f - 0.1f;
    hipMemcpy(in, h, 4096, hipMemcpyHostToDevice);
    run<4>(in, out, t, "one B set, ArchVGPR (control)");
    run<0>(in, out, t, "X Arch, Y Arch");
    run<1>(in, out, t, "X Arch, Y Acc");
    run<2>(in, out, t, "X Acc, Y Acc");
    run<3>(in, out, t, "X Acc, Y Acc, Y rewritten before tile 12");
    run<5>(in, out, t, "X Arch, Y Arch, Y gets NEW VALUES (from the accumulators) before tile 12");
    run<6>(in, out, t, "X Arch, Y Arch, Y gets new values before tiles 5, 6, 12, 13");
    return 0;
}
