// capi_todo.cpp — header entry points whose kernels are not built yet: they fail loudly.
#include <stdio.h>
#include "../../include/nfx.h"
extern "C" {
#define TODO(name) return NFX_ENOSUP
size_t nfx_mlp128_packed_bytes(int, int, int, int) { return 0; }
int nfx_mlp128_pack_weights(const float* const[5], const float* const[5], int, int, int, int, void*, size_t) { TODO(); }
int nfx_mlp128_xyz_fwd(const float*, int64_t, float, const void*, int, int, float, float, int, float*, void*) { TODO(); }
int nfx_lvis_fwd(const float*, int64_t, float, const float*, int, const void*, int, float*, void*) { TODO(); }
int nfx_shade_fwd(const float*, const float*, const float*, const float*, const float*, const float*, float, float,
                  const float*, const float*, const float*, const float*, int64_t, int, int, int, float*, void*) { TODO(); }
int nfx_shade_olat_fwd(const float*, const float*, const float*, const float*, const float*, const float*, float,
                       float, const float*, const float*, const float*, float, float, int64_t, int, int, float*,
                       void*) { TODO(); }
int nfx_brdf_spec_fwd(const float*, const float*, const float*, const float*, int, const float*, int, const void*,
                      int, int64_t, float*, void*) { TODO(); }
int nfx_dir2rusink(const float*, const float*, int64_t, float*, void*) { TODO(); }
}
