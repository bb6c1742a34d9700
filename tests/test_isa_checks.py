"""Checks on the gfx950 machine code of the windowed SpMM kernels (compiled here with the flags of
sparc_amd.build; no GPU needed):

* a DPP instruction must not read a register that a vector instruction wrote in the two instructions before it
  (gfx9 needs two wait states there; the broadcasts are compiler builtins so that the compiler provides them);
* the asynchronous global->LDS staging must be complete (s_waitcnt vmcnt(0)) before the workgroup barrier that
  publishes the window;
* no kernel of the per-step path spills more than a few registers to scratch memory.
"""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def spmm_asm(tmp_path_factory):
    from sparc_amd import build
    out = tmp_path_factory.mktemp("isa") / "spmm.s"
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = [f for f in build.FLAGS if f != "-fPIC"]
    subprocess.run([hipcc] + flags + ["-S", "--cuda-device-only", "-o", str(out),
                                      os.path.join(ROOT, "sparc_amd", "csrc", "spmm.hip")], check=True, cwd=str(out.parent))
    text = out.read_text()
    kernels = {}
    for m in re.finditer(r"^(_Z\w*spmm_win\w*):[^\n]*\n(.*?)^\.Lfunc_end", text, re.S | re.M):   # (a kernel may end in several places)
        body = [l.split(";")[0].strip() for l in m.group(2).splitlines()]     # drop comments
        kernels[m.group(1)] = [l for l in body if l and not l.startswith(".") and not l.endswith(":")]
    assert len(kernels) >= 16, "spmm_win instantiations not found in the assembly"
    return text, kernels


def _regs(operand):
    """Register numbers named by one operand (v12 or v[12:15])."""
    m = re.fullmatch(r"v(\d+)", operand)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", operand)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def dpp_hazards(ins):
    """(count of DPP instructions, list of (writer, reader, wait states) violations) of one instruction list."""
    n_dpp, bad = 0, []
    for i, line in enumerate(ins):
        op, _, rest = line.partition(" ")
        if not op.endswith("_dpp"):
            continue
        n_dpp += 1
        ops = [o.strip() for o in rest.split(" row_")[0].split(",")]
        src = _regs(ops[1])                       # the DPP operand is src0
        waited = 0
        for prev in reversed(ins[max(0, i - 4):i]):
            pop, _, prest = prev.partition(" ")
            if pop == "s_nop":
                waited += int(prest.strip()) + 1
            else:
                if pop.startswith("v_") and (_regs(prest.split(",")[0].strip()) & src):
                    bad.append((prev, line, waited))
                waited += 1
            if waited >= 2:
                break
    return n_dpp, bad


def test_no_dpp_read_within_two_wait_states_of_a_vector_write(spmm_asm):
    _, kernels = spmm_asm
    # the checker itself: one wait state is a violation, a vector write two instructions earlier or an s_nop 1 is not
    add = "v_add_u32_dpp v0, v18, v73 row_newbcast:0 row_mask:0xf bank_mask:0xf bound_ctrl:1"
    assert dpp_hazards(["v_lshlrev_b32 v18, 8, v44", "s_or_b64 exec, exec, s[4:5]", add])[1]
    assert dpp_hazards(["v_mov_b32 v18, v2", add])[1]
    assert not dpp_hazards(["v_mov_b32 v18, v2", "s_nop 1", add])[1]
    assert not dpp_hazards(["v_mov_b32 v18, v2", "v_mov_b32 v3, v2", "v_mov_b32 v4, v2", add])[1]
    total = 0
    for name, ins in kernels.items():
        n, bad = dpp_hazards(ins)
        total += n
        assert not bad, (name, bad[:3])
    assert total > 1000     # the gather of every instantiation is DPP based: the check saw it


def test_lds_staging_is_waited_for_before_the_barrier(spmm_asm):
    _, kernels = spmm_asm
    for name, ins in kernels.items():
        # (the pass that carries the column-sum finish holds that workgroup's barrier too: the barrier that publishes
        # the window is the first one after the staging copies)
        last_dma = max(i for i, l in enumerate(ins) if "global_load_lds" in l)
        bar = next(i for i, l in enumerate(ins) if i > last_dma and l.startswith("s_barrier"))
        between = ins[last_dma:bar]
        assert any(re.match(r"s_waitcnt vmcnt\(0\)", l) for l in between), name


def test_per_step_kernels_do_not_spill(spmm_asm):
    text, _ = spmm_asm
    spills = {}
    for m in re.finditer(r"\.name:\s+(\S*spmm_win\S*)\n(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)", text):
        spills[m.group(1)] = int(m.group(2))
    assert len(spills) >= 16
    assert max(spills.values()) <= 8, {k: v for k, v in spills.items() if v > 8}


def test_block_kernel_staging_and_registers(tmp_path):
    """dense_blocks (blocks.hip): the asynchronous global->LDS staging is waited for (vmcnt(0)) before the first
    synchronisation that publishes it (the barrier, or the staging counter of the fused variants), the matrix
    instruction is the 32x32x2 f32 one, and no variant spills."""
    from sparc_amd import build
    out = tmp_path / "blocks.s"
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = [f for f in build.FLAGS if f != "-fPIC"]
    subprocess.run([hipcc] + flags + ["-S", "--cuda-device-only", "-o", str(out),
                                      os.path.join(ROOT, "sparc_amd", "csrc", "blocks.hip")], check=True, cwd=str(tmp_path))
    text = out.read_text()
    found = 0
    for m in re.finditer(r"^(_Z\w*dense_blocks\w*):[^\n]*\n(.*?)^\.Lfunc_end", text, re.S | re.M):   # (the staging wait's trap ends the kernel too)
        ins = [l.split(";")[0].strip() for l in m.group(2).splitlines()]
        ins = [l for l in ins if l and not l.startswith(".") and not l.endswith(":")]
        found += 1
        last_dma = max(i for i, l in enumerate(ins) if "global_load_lds" in l)
        sync = next(i for i, l in enumerate(ins[last_dma:]) if l.startswith("s_barrier") or l.startswith("ds_add")) + last_dma
        assert any(re.match(r"s_waitcnt vmcnt\(0\)", l) for l in ins[last_dma:sync]), m.group(1)
        assert any("v_mfma_f32_32x32x2" in l for l in ins), m.group(1)
    assert found == 5     # plain / add, each alone or with the gathers, and the last pass with its loss epilogue
    spills = [int(x) for x in re.findall(r"\.name:\s+\S*dense_blocks\S*\n(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)", text)]
    assert len(spills) == 5 and max(spills) == 0, spills


def test_wide_dgrad_gemm_loop_waits_only_for_the_chunk_it_reads(tmp_path):
    """dgrad_wide_relu_kernel (mlp.hip): its global->LDS copies are issued two chunks ahead, so the product loop must
    wait for the copies of the chunk it is about to read and no further (vmcnt(6) with the next chunk's six copies in
    flight, vmcnt(0) only on the way out), synchronise with a bare s_barrier, and hold 64 matrix instructions per
    chunk; every instantiation fits the registers of two workgroups per CU without spilling, and the outputs of the narrow
    forward with the Gram product in its epilogue do not spill either."""
    from sparc_amd import build
    out = tmp_path / "mlp.s"
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = [f for f in build.FLAGS if f != "-fPIC"]
    subprocess.run([hipcc] + flags + ["-S", "--cuda-device-only", "-o", str(out),
                                      os.path.join(ROOT, "sparc_amd", "csrc", "mlp.hip")], check=True, cwd=str(tmp_path))
    text = out.read_text()
    metas = list(re.finditer(r"\.amdhsa_kernel (\S*dgrad_wide_relu_kernel\S*)(.*?)\.end_amdhsa_kernel", text, re.S))
    assert len(metas) == 5, "dgrad_wide_relu_kernel: the plain product and four widths of the first-layer form"
    for meta in metas:
        assert int(re.search(r"\.amdhsa_private_segment_fixed_size\s+(\d+)", meta.group(2)).group(1)) == 0, meta.group(1)
        assert int(re.search(r"\.amdhsa_next_free_vgpr\s+(\d+)", meta.group(2)).group(1)) <= 256, meta.group(1)
        assert int(re.search(r"\.amdhsa_group_segment_fixed_size\s+(\d+)", meta.group(2)).group(1)) <= 80 * 1024
    bodies = list(re.finditer(r"^(_Z\w*dgrad_wide_relu_kernel\w*):[^\n]*\n(.*?)^\.Lfunc_end", text, re.S | re.M))
    assert len(bodies) == 5
    for found in bodies:
        ins = [l.split(";")[0].strip() for l in found.group(2).splitlines()]
        ins = [l for l in ins if l and not l.startswith(".")]
        first_mfma = next(i for i, l in enumerate(ins) if "v_mfma_f32_32x32x2" in l)
        barrier = max(i for i, l in enumerate(ins[:first_mfma]) if l.startswith("s_barrier"))
        between = ins[barrier:first_mfma]
        assert sum("global_load_lds_dwordx4" in l for l in between) == 6
        assert not any(re.match(r"s_waitcnt.*vmcnt", l) for l in between), between
        before = [l for l in ins[max(0, barrier - 12):barrier] if l.startswith("s_waitcnt")]
        assert any("vmcnt(6)" in l for l in before) and not any("lgkmcnt(0)" in l and "vmcnt(0)" in l for l in before), before
        assert sum("v_mfma_f32_32x32x2" in l for l in ins) == 64
    for m in re.finditer(r"\.amdhsa_kernel (\S*linear_narrow_fwd_kernel\S*)(.*?)\.end_amdhsa_kernel", text, re.S):
        assert int(re.search(r"\.amdhsa_private_segment_fixed_size\s+(\d+)", m.group(2)).group(1)) == 0, m.group(1)
