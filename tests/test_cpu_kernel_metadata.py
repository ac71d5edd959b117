"""Static guard on the built code objects (no GPU): the kernels of lvis_v2.hip that run TWO waves per SIMD by construction
(`__launch_bounds__(512, 2)`: 512-thread workgroups, at most 256 registers) must not use scratch memory.  Round 6 (DESIGN.md
section 3.3): the one kernel of that file that fails bit identity with a partner wave on its SIMD, brdf_compact_kernel<2, 0, 8>
(an experiment build, not part of the product), is also the only one whose register allocation spills to scratch memory — and
the bisection showed scratch traffic to be an AMPLIFIER of its failure, not the cause (20 B per lane: 57 000 wrong elements
per 3e9 rows; none: 1 800; 32 B: 200 000), while the healthy two-wave kernels stay bit-identical even with scratch forced into
them.  The default light-visibility kernel resident128_kernel<2, 0, 8> and the opt-in brdf_compact_kernel<2, 1, 8> have no
scratch today; a change of compiler or source that makes them spill to memory fails here instead of on a customer's frame."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'scripts'))


@pytest.fixture(scope='module')
def rows(nfx_lib):
    import kernel_metadata
    if not os.path.exists(os.path.join(kernel_metadata.LLVM, 'llvm-objdump')):
        pytest.skip("no llvm-objdump")
    return kernel_metadata.kernels(os.path.join(ROOT, 'nerfactor_amd', 'libnfx.so'))


def test_every_kernel_has_metadata(rows):
    assert len(rows) >= 100
    assert all(r['vgpr_count'] is not None and r['max_flat_workgroup_size'] for r in rows)
    assert all(r['max_flat_workgroup_size'] <= 1024 for r in rows)


def test_two_wave_kernels_of_the_resident_networks_use_no_scratch(rows):
    two_wave = [r for r in rows if 'lv2::' in r['name'] and r['max_flat_workgroup_size'] == 512]
    names = sorted(r['name'].split('(')[0] for r in two_wave)
    assert any('resident128_kernel<2, 0, 8, false>' in n for n in names), names   # the default light-visibility kernel
    assert any('resident128_kernel<2, 0, 8, true>' in n for n in names), names    # ... and its form that stores at final rows (round 6)
    assert any('brdf_compact_kernel<2, 1, 8>' in n for n in names), names         # the opt-in learned-BRDF kernel
    assert not any('brdf_compact_kernel<2, 0, 8>' in n for n in names), "the failing form is not part of the product"
    for r in two_wave:
        assert r['private_segment_fixed_size'] == 0 and r['vgpr_count'] <= 256, r


def test_kernels_with_scratch_are_the_known_ones(rows):
    """Scratch is not forbidden (the fp32-class density-gradient kernel spills 82 dwords by design); a NEW kernel that starts
    to spill to memory should be a decision, not an accident."""
    known = ('nerf_sigma_x3_kernel<true>', 'nerf_mlp_bf16_rolled_kernel', 'nerf_mlp_bf16_kernel<1, 8>', 'nerf_mlp_bf16_v6_kernel<0, 2>',
             'nerf_mlp_bf16_v6_kernel<0, 0>', 'scatter_rows_kernel<HIP_vector_type<float, 4u> >', 'wgrad_lds_narrow_kernel', 'wgrad_lds_kernel')
    for r in rows:
        if r['private_segment_fixed_size']:
            assert any(k in r['name'] for k in known), (r['name'], r['private_segment_fixed_size'])
