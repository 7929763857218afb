"""Checks on the compiled gfx950 code objects inside libdgl_amd.so (no GPU needed: they are disassembled here).

The decode policy kernels split binary32 operands with a mixed-precision FMA written as inline assembly
(dgl_amd/csrc/split_f16.h).  Inline assembly is invisible to the compiler's hazard recognizer, so that split is only safe
where matrix results live in AGPRs, which the assembly's VGPR outputs can never alias.  A compiler or register-budget
change that moves those accumulators into VGPRs must not pass silently."""
import os
import re
import shutil
import subprocess

import pytest

LIB = os.path.join(os.path.dirname(__file__), "..", "dgl_amd", "lib", "libdgl_amd.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
# every function the split's inline assembly is compiled into
ASM_SPLIT_USERS = ("policy_kernel_f16ILi1ELi1EE", "policy_kernel_f16ILi2ELi1EE", "decode_policy_phase",
                   "policy_wide_kernel_f16ILi2EE", "policy_wide_kernel_f16ILi4EE")


@pytest.fixture(scope="module")
def disassembly(tmp_path_factory):
    if not os.path.exists(OBJDUMP) or not os.path.exists(LIB):
        pytest.skip("llvm-objdump or the library is not here")
    d = tmp_path_factory.mktemp("codeobj")
    lib = shutil.copy(LIB, d / "lib.so")
    subprocess.run([OBJDUMP, "--offloading", lib], check=True, capture_output=True)
    funcs = {}
    for f in sorted(os.listdir(d)):
        if "amdgcn" not in f:
            continue
        text = subprocess.run([OBJDUMP, "-d", str(d / f)], check=True, capture_output=True, text=True).stdout
        name = None
        for line in text.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                name = m.group(1)
                funcs.setdefault(name, [])
            elif name is not None:
                funcs[name].append(line)
    assert funcs, "no device code found in the library"
    return funcs


@pytest.mark.parametrize("user", ASM_SPLIT_USERS)
def test_inline_asm_split_only_beside_agpr_accumulators(disassembly, user):
    names = [n for n in disassembly if user in n]
    assert names, f"{user}: not in the library"
    for n in names:
        body = disassembly[n]
        mfma = [l for l in body if "v_mfma_" in l]
        mix = [l for l in body if "v_fma_mix_f32" in l]
        assert mfma and mix, f"{n}: expected matrix instructions and the split's mixed-precision FMAs"
        dst = [l.split("v_mfma_")[1].split()[1] for l in mfma]
        in_vgpr = [d for d in dst if not d.startswith("a")]
        assert not in_vgpr, (f"{n}: {len(in_vgpr)} of {len(mfma)} matrix instructions write VGPRs; build this function "
                             f"with DGL_SPLIT_NO_ASM (see split_f16.h)")


def test_vgpr_form_kernels_do_not_use_the_inline_asm(disassembly):
    """The converse: wherever a function's matrix instructions write VGPRs, the split must be the plain one."""
    for n, body in disassembly.items():
        mfma_v = [l for l in body if "v_mfma_" in l and not l.split("v_mfma_")[1].split()[1].startswith("a")]
        if mfma_v:
            assert not any("v_fma_mix_f32" in l for l in body), f"{n}: VGPR-form matrix results beside the inline-assembly split"


def test_wide_policy_kernels_fit_the_instruction_cache_and_do_not_spill(disassembly):
    """The kernels for rows of 33..128 tokens (policy16_wide.h): one block of the network is a loop body that has to fit
    the 64 KB instruction cache (as unrolled code it did not, and every pass was fetched from memory), and its products
    are pinned so that the register allocator has no operand fragments of several products to keep alive - a build
    that spills to scratch memory stalls a wave that has its SIMD to itself for a full memory round trip per reload."""
    for nt, spill_limit in ((2, 0), (4, 8)):
        names = [n for n in disassembly if f"policy_wide_kernel_f16ILi{nt}EE" in n]
        assert names
        body = disassembly[names[0]]
        insts = [l for l in body if re.match(r"^\s+[sv]_|^\s+(ds|global|scratch|buffer)_", l)]
        scratch = [l for l in body if "scratch_" in l]
        assert len(scratch) <= spill_limit, f"{names[0]}: {len(scratch)} scratch accesses"
        assert len(insts) * 8 < 80 * 1024, f"{names[0]}: about {len(insts) * 8 // 1024} KB of code"
