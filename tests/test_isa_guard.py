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
