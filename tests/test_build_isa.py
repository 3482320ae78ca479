"""CPU: properties of the generated gfx950 code that the hand-scheduled GEMM kernels rely on and that a compiler
change could silently break (checked on the device assembly, no GPU needed)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "visitron_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"

pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")


def _device_asm(src, tmp_path):
    out = os.path.join(str(tmp_path), os.path.basename(src) + ".s")
    subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", src, "-o", out],
                   check=True, cwd=CSRC, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return open(out).read()


def _kernels(asm):
    """name -> text of each kernel (label .. .end_amdhsa_kernel)."""
    res = {}
    for m in re.finditer(r"^(_Z\w+):.*?\.end_amdhsa_kernel", asm, re.S | re.M):
        res[m.group(1)] = m.group(0)
    return res


def test_agpr_gemm_kernels_have_no_scratch(tmp_path):
    """The 256x256-tile kernels keep 256 accumulators in AGPRs through inline-asm MFMAs; an accumulator array that
    falls to scratch (a loop left rolled, a dynamic index) is both slow and WRONG (the asm MFMAs are asynchronous)."""
    ks = _kernels(_device_asm(os.path.join(CSRC, "gemm_v7.hip"), tmp_path))
    assert len(ks) >= 30
    for name, text in ks.items():
        m = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", text)
        assert m and int(m.group(1)) == 0, name
        assert "scratch_" not in text, name


def test_deferred_layernorm_gemm_kernels_keep_their_ring_in_place(tmp_path):
    """gemm_v7_ln.hip: the deferred-LayerNorm epilogues.  No scratch; no register parked in an AGPR (v_accvgpr_write) --
    the residual-stream ring is written by loads hipcc does not see, and a copy made before the data has landed carries
    the old bits (that is what broke the 224- and 256-row tiles while 16 row factors and 16 running sums sat in registers
    beside the ring); and no EMPTY inline-asm statement: hipcc's hazard recognizer counts one as an instruction and drops
    the wait state a packed fp32 op needs before a dependent one (the pins carry an s_nop)."""
    ks = _kernels(_device_asm(os.path.join(CSRC, "gemm_v7_ln.hip"), tmp_path))
    assert len(ks) == 24                       # 8 tile heights x (2 activations of mode 1 + mode 2)
    for name, text in ks.items():
        m = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", text)
        assert m and int(m.group(1)) == 0, name
        assert "scratch_" not in text, name
        lines = [l.strip() for l in text.splitlines()]
        # (round 6: the stream-K fix-up of the 160- / 128-row persistent kernels -- between its first fp32 identity MFMA and
        # its last store of accumulators as a partial tile -- may move ACCUMULATORS between register files: those are
        # results of MFMAs behind the K loop's closing s_nops, not data of loads hipcc cannot see; the ring is not live there)
        sk = [i for i, l in enumerate(lines) if l.startswith("v_mfma_f32_16x16x4_f32") or (l.startswith("buffer_store_dwordx4 a[") and l.endswith("sc1"))]
        for i, l in enumerate(lines):
            if l.startswith("v_accvgpr_write"):
                assert sk and min(sk) - 400 <= i <= max(sk), (name, i, l)
        last_mfma = max(i for i, l in enumerate(lines) if l.startswith("v_mfma"))   # the epilogue follows the K loop's last MFMA
        for i in range(last_mfma, len(lines) - 1):
            assert not (lines[i].startswith(";;#ASMSTART") and lines[i + 1].startswith(";;#ASMEND")), (name, i)


def test_wgrad_v8_fragment_registers_are_left_alone(tmp_path):
    """gemm_wgrad_tn_v8 names v[96:227] in its instruction text (transposing-read fragments, the bias ones operand);
    nothing the compiler emits may touch them, and the code object's register count must cover them."""
    ks = _kernels(_device_asm(os.path.join(CSRC, "gemm_wgrad_v8.hip"), tmp_path))
    text = [t for n, t in ks.items() if "gemm_wgrad_tn_v8" in n][0]
    assert int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", text).group(1)) == 0
    assert int(re.search(r"\.amdhsa_accum_offset (\d+)", text).group(1)) >= 228
    assert int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", text).group(1)) >= 228 + 256
    for line in text.splitlines():
        t = line.strip()
        if not t or t[0] in ";." or t.startswith("ds_read_b64_tr_b16") or t.startswith("v_mfma"):
            continue
        regs = set()
        for m in re.finditer(r"\bv\[(\d+):(\d+)\]", t):
            regs.update(range(int(m.group(1)), int(m.group(2)) + 1))
        for m in re.finditer(r"\bv(\d+)\b", t):
            regs.add(int(m.group(1)))
        hit = [r for r in regs if 96 <= r <= 227]
        if hit:
            assert t.startswith("v_mov_b32 v22") and "0x3f803f80" in t, t   # the ones operand, set once at entry


def test_generated_wgrad_step_is_current():
    """visitron_amd/csrc/wgrad_v8_step.inc is generated (tools/gen_wgrad_step.py): the committed file must be what the
    generator prints."""
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "gen_wgrad_step.py")], check=True,
                         stdout=subprocess.PIPE).stdout.decode()
    assert out == open(os.path.join(CSRC, "wgrad_v8_step.inc")).read()


def test_attention_kernels_register_budgets_and_the_writelane_hazard(tmp_path):
    """attention_fwd.hip / attention_bwd.hip: the forward must stay within 128 VGPRs (two 8-wave workgroups per CU, four
    waves per SIMD: the kernel is bound by vector issue and lives on that occupancy), the inference instantiation without
    scratch; the backward without scratch.  The training forward moves each compare's lane mask into the lane of its key
    with v_writelane_b32 from inline asm: hipcc's hazard recognizer does not see into the statement, and without wait states
    after the v_cmp that wrote the SGPR the first word of a group carried the PREVIOUS compare's mask (measured) -- every
    group of v_writelane_b32 must follow an s_nop of at least 4."""
    asm = _device_asm(os.path.join(CSRC, "attention_fwd.hip"), tmp_path)
    ks = _kernels(asm)
    fwd = {n: t for n, t in ks.items() if "attention_fwd_d64" in n}
    assert len(fwd) == 8            # (keep words | none) x (per-query bias | none) x (8-bit | 16-bit dropout fields: round 6)
    for name, text in fwd.items():
        assert int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", text).group(1)) <= 128, name
        keep = "ILb1E" in name
        if not keep:
            assert int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", text).group(1)) == 0, name
            assert "v_writelane_b32" not in text, name
        else:
            lines = [l.strip() for l in text.splitlines() if l.strip() and not l.strip().startswith(";")]
            idx = [i for i, l in enumerate(lines) if l.startswith("v_writelane_b32")]
            assert len(idx) == 32, (name, len(idx))            # 16 compares x 2 half-waves per 32-key tile, in two groups of 16
            for i in idx:
                j = i
                while lines[j].startswith("v_writelane_b32"):
                    j -= 1
                m = re.match(r"s_nop (\d+)", lines[j])
                assert m and int(m.group(1)) >= 4, (name, lines[j])
    ks = _kernels(_device_asm(os.path.join(CSRC, "attention_bwd.hip"), tmp_path))
    bwd = {n: t for n, t in ks.items() if "attention_bwd_d64" in n}
    # 4 waves; 8 waves x (hash | keep words) x (delta pass | in-kernel); 16 waves and 16 waves persistent, each x (keep words | none)
    assert len(bwd) == 9
    for name, text in bwd.items():
        assert int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", text).group(1)) == 0, name
        if "w16" in name:       # sixteen waves per workgroup = four per SIMD: 128 registers is the whole budget
            assert int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", text).group(1)) <= 128, name
        if "w16p" in name:
            # the persistent kernel hides its loads behind LDS-DMA issued from inline asm; between a pair's first and last
            # iteration no compiler-visible global load may appear (hipcc waits for each with a vmcnt that also drains the
            # DMA pieces in flight): the only plain loads are the V fragments / mask value (prologue and each pair's last step)
            loads = [l for l in text.splitlines() if re.match(r"\s*global_load_dword", l) and "lds" not in l]
            assert len(loads) <= 8, (name, len(loads))
