"""CPU-only guard for every kernel that hand-counts `s_waitcnt` behind inline-assembly loads:
the sources are compiled to gfx950 assembly here (hipcc cross-compiles without a GPU) and walked by tests/isa_lint.py.

What the guard is for: hipcc does not know that the destination of an inline-assembly `ds_read_*` / `global_load_*` is invalid
until the wait the source counts by hand; a register copy, spill or hoisted use in front of that wait passes every op-level test
and shows up as run-to-run differences of the train step (round 4: `wgradr`; round 5: this guard found the same pattern in the
run-time fragment loop of `stemp_kernel<0>`, see csrc/stemp.hip)."""
import re
import shutil
import subprocess
import tempfile
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import pytest

import isa_lint

ROOT = Path(__file__).resolve().parents[1]
CSRC = ROOT / "nkb-classification_amd" / "csrc"
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

# file -> (kernel name pattern, minimum number of inline-assembly loads the walk must have seen: the parser did not go blind)
STRICT = {
    "convp.hip": ("convp", 2000),
    "stemp.hip": ("stemp", 500),
    "gramr.hip": ("gramr_kernel", 40),
    "wgrad256.hip": ("wgrad8", 90),
    "wgrad3x3.hip": ("wgrad3x3p_kernel", 250),
    "wgradr.hip": ("wgradr_kernel", 110),
    "gemm8p.hip": ("gemm8p_kernel", 150),
}
# conv1p.hip waits with RUN-TIME counts (a switch over `s_waitcnt vmcnt(k)`, c1_vmcnt_dyn): a path-insensitive walk joins the
# arms and reports fragments as in flight on paths that cannot happen.  Held to the compiler-copy criterion instead: no v_mov /
# v_accvgpr / v_swap / scratch instruction may READ a register the walk considers in flight.
COPIES = {"conv1p.hip": ("conv1[ps]_kernel", 1500)}
_COPY_OPS = ("v_mov_", "v_accvgpr_", "v_swap_", "scratch_", "v_permlane", "ds_write", "ds_bpermute", "v_readlane", "v_readfirstlane")

_cache = {}


def _assembly(name: str) -> str:
    if name not in _cache:
        if not Path(HIPCC).exists():
            pytest.skip("hipcc not available")
        names = [n for n in list(STRICT) + list(COPIES) if n not in _cache]

        def build(n):
            with tempfile.TemporaryDirectory() as td:
                out = Path(td) / "k.s"
                subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", f"-I{CSRC}", f"-I{ROOT / 'include'}",
                                "-ffp-contract=off", "-fno-slp-vectorize", "-S", "--cuda-device-only", "-o", str(out), str(CSRC / n)],
                               check=True, capture_output=True, timeout=900)
                return n, out.read_text()

        with ThreadPoolExecutor(max_workers=4) as pool:      # (the flags are csrc/Makefile's: the guard sees the shipped code)
            for n, text in pool.map(build, names):
                _cache[n] = text
    return _cache[name]


def _sources(f: isa_lint.Finding):
    code = f.text.split(";")[0].strip()
    op = code.split()[0]
    ops = code[len(op):].split(",")
    is_store = op.startswith(("ds_write", "global_store", "buffer_store", "scratch_store"))
    return op, isa_lint._regs(",".join(ops if is_store else ops[1:]))


def test_makefile_flags_are_the_ones_the_guard_compiles_with():
    mk = (CSRC / "Makefile").read_text()
    for flag in ("-O3", "-ffp-contract=off", "-fno-slp-vectorize"):
        assert flag in mk, flag


def test_lint_flags_a_copy_in_front_of_its_counted_wait_and_accepts_the_waited_form():
    """positive and negative control on a hand-written fragment (the round-4 `wgradr` pattern: a phi copy at the top of a stage)"""
    bad = """
_Z9bad_kernelv:                         ; @bad
; %bb.0:
	v_mov_b32_e32 v9, 0
.LBB0_1:                                ; =>This Inner Loop Header: Depth=1
	v_mov_b64_e32 v[20:21], v[30:31]
	;;#ASMSTART
	s_waitcnt lgkmcnt(0)
	;;#ASMEND
	v_mfma_f32_16x16x32_bf16 v[0:3], v[20:23], v[40:43], v[0:3]
	;;#ASMSTART
	ds_read_b64_tr_b16 v[30:31], v9 offset:0
	;;#ASMEND
	;;#ASMSTART
	ds_read_b64_tr_b16 v[32:33], v9 offset:512
	;;#ASMEND
	s_add_i32 s4, s4, 1
	s_cmp_lt_i32 s4, s5
	s_cbranch_scc1 .LBB0_1
; %bb.2:
	s_waitcnt lgkmcnt(0)
	s_endpgm
.Lfunc_end0:
"""
    findings, seen = isa_lint.lint(bad, "bad_kernel")
    assert seen == {"_Z9bad_kernelv": 2}
    assert [f.text.strip() for f in findings] == ["v_mov_b64_e32 v[20:21], v[30:31]"], [str(f) for f in findings]
    good = bad.replace("\tv_mov_b64_e32 v[20:21], v[30:31]\n\t;;#ASMSTART\n\ts_waitcnt lgkmcnt(0)\n\t;;#ASMEND\n",
                       "\t;;#ASMSTART\n\ts_waitcnt lgkmcnt(0)\n\t;;#ASMEND\n\tv_mov_b64_e32 v[20:21], v[30:31]\n")
    assert good != bad
    findings, _ = isa_lint.lint(good, "bad_kernel")
    assert findings == [], [str(f) for f in findings]
    # an unconditional branch ends the fall-through: the block behind it is reached only through its label
    skip = bad.replace("; %bb.2:\n", "\ts_branch .LBB0_3\n.LBB0_9:\n\tv_mov_b32_e32 v1, v30\n.LBB0_3:\n")
    findings, _ = isa_lint.lint(skip.replace("\tv_mov_b64_e32 v[20:21], v[30:31]\n", ""), "bad_kernel")
    assert findings == [], [str(f) for f in findings]


@pytest.mark.parametrize("src", sorted(STRICT))
def test_no_instruction_touches_a_register_with_an_assembly_load_in_flight(src):
    pattern, least = STRICT[src]
    findings, seen = isa_lint.lint(_assembly(src), pattern)
    assert seen and sum(seen.values()) >= least, (src, seen)
    assert not findings, f"{src}: {len(findings)} findings, first: " + "; ".join(str(f) for f in findings[:5])


@pytest.mark.parametrize("src", sorted(COPIES))
def test_no_compiler_copy_reads_a_register_with_an_assembly_load_in_flight(src):
    pattern, least = COPIES[src]
    findings, seen = isa_lint.lint(_assembly(src), pattern)
    assert seen and sum(seen.values()) >= least, (src, seen)
    copies = []
    for f in findings:
        op, srcs = _sources(f)
        if op.startswith(_COPY_OPS) and srcs & set(f.regs):
            copies.append(f)
    assert not copies, f"{src}: " + "; ".join(str(f) for f in copies[:5])


def test_weight_gradient_pipelines_keep_their_main_loops_free_of_register_traffic():
    """the round-4 form of the guard (kept: it also forbids copies of registers that are NOT in flight inside the two pipelines'
    main loops — 204 v_accvgpr moves per k-step was what `__launch_bounds__(256, 2)` removed)"""
    for src, kernels, loop_index in (("wgrad3x3.hip", "wgrad3x3p_kernel", 2), ("wgradr.hip", "wgradr_kernel", 1)):
        text = _assembly(src)
        bodies = re.findall(r"\n(_Z\w*%s\w*):[^\n]*\n(.*?)s_endpgm" % kernels, text, flags=re.S)
        assert bodies, src
        for name, body in bodies:
            lines = body.split("\n")
            heads = [i for i, l in enumerate(lines) if "Loop Header" in l]
            assert len(heads) >= loop_index, (name, len(heads))
            start = heads[loop_index - 1]
            end = next(i for i in range(start, len(lines)) if "global_store" in lines[i] or "global_atomic" in lines[i])
            loop = lines[start:end]
            assert sum("v_mfma" in l for l in loop) >= 32 and sum("ds_read_b64_tr_b16" in l for l in loop) >= 24, name
            assert not [l for l in loop if "v_mov_b64" in l or "v_accvgpr" in l or "scratch_" in l], name
            movs = {l.split(",")[-1].strip() for l in loop if "v_mov_b32" in l}
            assert len(movs) <= 1, (name, sorted(movs))


def test_assembly_memory_instructions_wait_for_valu_written_scalar_operands():
    """gemm8p.hip is the one source whose inline-assembly loads take SCALAR operands (saddr bases, M0 for the LDS-DMA): each must
    stand 5 wait states behind a VALU write of that SGPR (v_readlane out of an SGPR spill) — see isa_lint.lint_vmem_sgpr_hazard."""
    bad = """
_Z9bad_kernelv:
	v_readlane_b32 s12, v255, 13
	v_readlane_b32 s13, v255, 14
	v_add_u32_e32 v0, s18, v215
	;;#ASMSTART
	global_load_dwordx4 v[0:3], v0, s[12:13]
	;;#ASMEND
	s_endpgm
.Lfunc_end0:
"""
    findings, seen = isa_lint.lint_vmem_sgpr_hazard(bad, "bad_kernel")
    assert seen == 1 and len(findings) == 1
    findings, _ = isa_lint.lint_vmem_sgpr_hazard(bad.replace("\tglobal_load", "\ts_nop 4\n\tglobal_load"), "bad_kernel")
    assert findings == []
    findings, seen = isa_lint.lint_vmem_sgpr_hazard(_assembly("gemm8p.hip"), "gemm8p_kernel")
    assert seen >= 100, seen
    assert not findings, "; ".join(str(f) for f in findings[:5])
    for src in sorted(set(STRICT) | set(COPIES)):                     # nobody else feeds scalar operands to assembly loads today
        if src != "gemm8p.hip":
            f2, _ = isa_lint.lint_vmem_sgpr_hazard(_assembly(src), STRICT.get(src, COPIES.get(src))[0])
            assert not f2, (src, str(f2[0]))


def test_persistent_gemm_loops_never_drain_the_dma_stream():
    """Round 5: a compiler-VISIBLE load anywhere in gemm8p's persistent kernels lets hipcc's wait-count pass put `s_waitcnt vmcnt(0)`
    where it believes a result register is overwritten — twice that was the top of the k-loop (the whole LDS-DMA stream drained once
    per k-tile: +8 % on every launch; the fp8 kernels had carried one since they were written).  The bf16 / fp8 kernels without the
    quantised second output must hold no compiler-visible global load at all, no spill traffic inside the k-loop, and exactly ONE wait on
    the vector-memory counter there: the counted vmcnt(6) of phase 4."""
    text = _assembly("gemm8p.hip")
    found = 0
    for m in re.finditer(r"\n(_Z\w*gemm8p_kernelILb1ELi[012]ELb0E\w*):[^\n]*\n(.*?)\n\.Lfunc_end\d+:", text, flags=re.S):
        found += 1
        lines = m.group(2).split("\n")
        in_asm, visible = False, []
        for l in lines:
            if ";;#ASMSTART" in l:
                in_asm = True
            elif ";;#ASMEND" in l:
                in_asm = False
            elif not in_asm and re.match(r"\s*(global_load|buffer_load|flat_load)", l):
                visible.append(l.strip())
        assert not visible, (m.group(1), visible[:3])
        mf = [i for i, l in enumerate(lines) if "v_mfma" in l]
        head = max(i for i in range(mf[0]) if "Loop Header" in lines[i])
        loop = lines[head:mf[-1]]
        # (a spilled register may be reloaded in an epilogue; inside the k-loop a reload — or any wait the pass derives from one — is the drain)
        assert not [l for l in loop if "scratch_" in l], m.group(1)
        waits = [l.strip() for l in loop if "vmcnt" in l]
        assert waits == ["s_waitcnt vmcnt(6)"], (m.group(1), waits)
    assert found == 3
    # the kernels with the quantised second output spill around their epilogues; a wait the compiler can see, once per tile behind the
    # epilogue, keeps the reloads' waits out of the k-loop (they had been vmcnt(4) + vmcnt(0) at its top: +15 % on a K = 4 096 launch)
    found = 0
    for m in re.finditer(r"\n(_Z\w*gemm8p_kernelILb1ELi[12]ELb1E\w*):[^\n]*\n(.*?)\n\.Lfunc_end\d+:", text, flags=re.S):
        found += 1
        lines = m.group(2).split("\n")
        mf = [i for i, l in enumerate(lines) if "v_mfma" in l]
        head = max(i for i in range(mf[0]) if "Loop Header" in lines[i])
        loop = lines[head:mf[-1]]
        assert not [l for l in loop if "scratch_" in l], m.group(1)
        assert [l.strip() for l in loop if "vmcnt" in l] == ["s_waitcnt vmcnt(6)"], m.group(1)
    assert found == 2
