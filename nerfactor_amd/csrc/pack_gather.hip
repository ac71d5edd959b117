// pack_gather.hip — device-side re-packing of a weight blob after an optimizer step.
// Every blob of pack.cpp is a pure gather of the network's parameters (bf16 fragments in MFMA lane order + fp32
// biases, zero padding), so a precomputed index map turns "pack" into one coalesced pass on the GPU: no
// device->host copy, no host packer, no upload inside a training step (the map is built once per network by
// nerfactor_amd/ops.py:DevicePacker from the host packer itself).
#include "nfx_common.hpp"

namespace nfx {
// One thread per 32-bit word of the blob; map[2i], map[2i+1]:
//   (a, -2)  -> the word is fp32 src[a]            (a < 0: 0.0f)
//   (a,  b)  -> the word is the bf16 pair {src[a], src[b]} (low half first; negative index: 0).  An index with bit 30 set
//               (round 5) asks for the LO half of the fp32-class operand pair of mlp_x3.hpp instead: bf16(v - bf16(v)) —
//               the split hi / lo blobs of the tuned fp32-class kernels are gathers of (hi, lo) halves, so they too are
//               re-packed on the device after an optimizer step (precision = fp32 training re-packed them on the host).
__global__ void pack_gather_kernel(const float* __restrict__ src, const int2* __restrict__ map, long long n_words,
                                   unsigned* __restrict__ blob) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_words) return;
    const int2 m = map[i];
    unsigned out;
    if (m.y == -2) {
        out = m.x >= 0 ? __float_as_uint(src[m.x]) : 0u;
    } else {
        auto half = [&](int idx) -> __bf16 {
            if (idx < 0) return (__bf16)0.f;
            const float v = src[idx & 0x3fffffff];
            const __bf16 h = (__bf16)v;
            return (idx & 0x40000000) ? (__bf16)(v - (float)h) : h;
        };
        const __bf16 lo = half(m.x), hi = half(m.y);
        out = (unsigned)__builtin_bit_cast(unsigned short, lo) | ((unsigned)__builtin_bit_cast(unsigned short, hi) << 16);
    }
    blob[i] = out;
}
}  // namespace nfx

extern "C" int nfx_launch_pack_gather(const float* src, const int* map, long long n_words, void* blob, hipStream_t st) {
    if (n_words <= 0) return 0;
    hipLaunchKernelGGL(nfx::pack_gather_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, st, src,
                       reinterpret_cast<const int2*>(map), n_words, static_cast<unsigned*>(blob));
    return (int)hipGetLastError();
}
