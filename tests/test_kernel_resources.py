"""What the march's speed stands on besides its arithmetic, checked on the build's own ISA listing (no GPU): the register budgets
that decide how many waves a SIMD holds, no scratch memory in the lean kernels, and no packed fp32 instruction anywhere (round 5
measured v_pk_fma_f32 / v_pk_add_f32 2-5 % slower than the scalar instructions they replace: profiles/r05_ab_step_asm.txt).
The listing is what csrc/build.sh keeps from -save-temps in $VRT_BUILD_TMP (/tmp/vrtbuild); where the library was built
elsewhere (the GPU box gets the .so, not the listing) the tests skip."""
import os
import re

import pytest

LISTING = os.path.join(os.environ.get("VRT_BUILD_TMP", "/tmp/vrtbuild"), "vrt_kernels-hip-amdgcn-amd-amdhsa-gfx950.s")
LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "volumetricraytracer_amd", "lib", "libvrt_hip.so")


def _listing():
    if not os.path.exists(LISTING) or not os.path.exists(LIB) or os.path.getmtime(LISTING) + 600 < os.path.getmtime(LIB):
        pytest.skip("no ISA listing of this build here (it is written by csrc/build.sh next to the build's temporaries)")
    return open(LISTING).read()


def _kernels(text):
    meta = text[text.index("amdhsa.kernels:"):]
    out = {}
    for block in re.split(r"\n  - \.agpr_count:", meta)[1:]:
        f = dict(re.findall(r"\.(name|vgpr_count|sgpr_count|private_segment_fixed_size|vgpr_spill_count):\s+(\S+)", block))
        out[f["name"]] = {k: int(v) for k, v in f.items() if k != "name"}
    return out


def _march_kernels(k):
    """{(PATH, SINGLE, DIAG, DYN, REF): resources} of the march_kernel<PATH, SINGLE, DIAG, DYN, REF> instantiations (Itanium names)."""
    out = {}
    for name, r in k.items():
        m = re.fullmatch(r"_ZN3vrt12march_kernelILi(\d+)ELb([01])ELb([01])ELb([01])ELb([01])EEEvNS_6DBlockE", name)
        if m:
            out[(int(m.group(1)),) + tuple(x == "1" for x in m.groups()[1:])] = r
    return out


def test_the_lean_kernels_fit_eight_waves_per_simd_and_use_no_scratch():
    mk = _march_kernels(_kernels(_listing()))
    # SINGLE, not the diagnostic build, every data path but the dense grid's (path 1: no tables, a debugging path at 66 registers)
    lean = {t: r for t, r in mk.items() if t[1] and not t[2] and t[0] != 1}
    assert len(lean) >= 16, sorted(mk)
    for t, r in lean.items():
        # 64 registers = 8 waves per SIMD (the hardware's cap, and what the march's latency hiding is sized for)
        assert r["vgpr_count"] <= 64, (t, r)
        assert r["private_segment_fixed_size"] == 0 and r["vgpr_spill_count"] == 0, (t, r)
    # the instantiation the benchmark runs (fp32 bricks, no reference switches) keeps some head room
    assert lean[(2, True, False, False, False)]["vgpr_count"] <= 60, lean[(2, True, False, False, False)]


def test_the_bvh_kernels_fit_seven_waves_per_simd():
    mk = _march_kernels(_kernels(_listing()))
    bvh = {t: r for t, r in mk.items() if t[0] != 1 and not t[1] and not t[2]}
    assert len(bvh) >= 8, sorted(mk)
    for t, r in bvh.items():
        assert r["vgpr_count"] <= 72, (t, r)
    # the plain one (config 5) has no spills since it reads the table word in every trip
    assert bvh[(2, False, False, False, False)]["vgpr_spill_count"] == 0, bvh[(2, False, False, False, False)]


def test_no_packed_fp32_instruction_in_the_build():
    text = _listing()
    code = text[:text.index("amdhsa.kernels:")]
    packed = sorted(set(re.findall(r"^\s+(v_pk_(?:fma|add|mul)_f32)\b", code, flags=re.M)))
    assert packed == [], packed


def test_the_hand_written_step_blocks_are_in_the_march_loops():
    """step_from_sample / step_over_empty_space: one asm block each per march loop of the lean kernel (camera ray, shadow ray), with the
    scalar condition code among the clobbers (a block with s_and/s_andn2 and without it lets the compiler keep a loop's s_cmp result
    across it: tools/microbench/issue_cost.hip hung on that)."""
    text = _listing()
    m = re.search(r"^_ZN3vrt12march_kernelILi2ELb1ELb0ELb0ELb0EEEvNS_6DBlockE:[^\n]*\n(.*?)\n\.Lfunc_end", text, flags=re.S | re.M)
    assert m
    body = m.group(1)
    assert body.count("v_cmpx_gt_f32_e32") >= 2 and body.count("s_and_saveexec_b64") >= 2
    src = open(os.path.join(os.path.dirname(LIB), "..", "csrc", "vrt_kernels.hip")).read()
    block = src[src.index("void step_from_sample("):src.index("void step_over_empty_space(")]
    assert re.search(r':\s*"vcc",\s*"scc"\);', block), "step_from_sample must declare vcc and scc clobbered"
