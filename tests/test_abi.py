"""CPU tests of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports exactly what
include/effort_hip.h declares (no compute calls -- there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions(names=("effort_hip.h", "effort_hip_debug.h")):
    fns = set()
    for n in names:
        src = open(os.path.join(ROOT, "include", n)).read()
        fns |= set(re.findall(r"EFFORT_API\s+[\w\s\*]+?\b(effort_\w+)\s*\(", src))
    return sorted(fns)


def test_header_declares_the_expected_surface():
    fns = _header_functions(("effort_hip.h",))
    # the lab bench (profiling / tracing / ablation hooks) is not part of the drop-in boundary
    assert not [f for f in fns if f.startswith(("effort_debug_", "effort_kernel_", "effort_enable_kernel_timing", "effort_set_persistent",
                                                 "effort_set_tuning", "effort_set_split_cutoff"))]
    for must in ("effort_create", "effort_destroy", "effort_sync", "effort_weights_fp16", "effort_weights_q4",
                 "effort_bucketmul", "effort_bucketmul_q4", "effort_dense_gemv", "effort_convert_fp16",
                 "effort_last_dispatch_count", "effort_calc_dispatch"):
        assert must in fns


def test_headers_compile_as_c11():
    """A Swift / cgo / JNI binding imports the header as C: it must be C on its own, not only when included from .hip."""
    import subprocess
    for n in ("effort_hip.h", "effort_hip_debug.h"):
        subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-x", "c", os.path.join(ROOT, "include", n)])


def _prototypes():
    """{name: (return type, [parameter types])} parsed from the headers, types reduced to what an ABI sees: every pointer is
    'ptr', scalars keep their C type."""
    protos = {}
    for n in ("effort_hip.h", "effort_hip_debug.h"):
        src = open(os.path.join(ROOT, "include", n)).read()
        src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)                       # comments may hold parentheses and commas
        for ret, name, params in re.findall(r"EFFORT_API\s+([\w\s\*]+?)\b(effort_\w+)\s*\(([^)]*)\)\s*;", src):
            def kind(t):
                t = t.strip()
                if "*" in t:
                    return "ptr"
                words = [w for w in t.split() if w not in ("const", "unsigned")]
                return ("u" if "unsigned" in t.split() else "") + words[0]
            plist = [] if params.strip() in ("", "void") else [kind(re.sub(r"\b\w+\s*$", "", p) if not p.strip().endswith("*") else p) for p in params.split(",")]
            protos[name] = (kind(ret), plist)
    return protos


def test_python_binding_matches_the_prototypes():
    """Argument COUNT and C TYPES of every entry of effort_amd._lib._SIGS against the header's prototypes: a drifted
    `int` vs `int64_t` or a missing parameter would pass the name check and corrupt the stack at run time."""
    import ctypes as C

    from effort_amd import _lib
    protos = _prototypes()
    assert sorted(protos) == sorted(_lib._SIGS)

    def ckind(t):
        if t is None:
            return "void"
        if t in (C.c_void_p, C.c_char_p) or isinstance(t, type(C.POINTER(C.c_int))):
            return "ptr"
        return {C.c_int: "int", C.c_int64: "int64_t", C.c_double: "double", C.c_float: "float", C.c_uint32: "uint32_t"}[t]

    for name, (res, args) in _lib._SIGS.items():
        ret, params = protos[name]
        assert ckind(res) == ret, (name, "return", ckind(res), ret)
        assert [ckind(a) for a in args] == params, (name, [ckind(a) for a in args], params)


def test_library_loads_and_exports_every_declared_symbol(hip_lib_built):
    lib = ctypes.CDLL(hip_lib_built)
    for name in _header_functions():
        assert hasattr(lib, name), f"{name} declared in effort_hip.h but not exported"
    lib.effort_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.effort_version()


def test_python_binding_covers_the_header(hip_lib_built):
    import effort_amd
    assert effort_amd._lib.exported_symbols() == _header_functions()
    effort_amd.lib()        # binds every prototype; raises if one is missing


def test_null_arguments_are_reported_not_crashed(hip_lib_built):
    import effort_amd
    lib = effort_amd.lib()
    assert lib.effort_sync(None) == -1
    assert lib.effort_bucketmul(None, None, None, None, None, 0.25) == -1
    assert lib.effort_set_tuning(None, 0, 0, 0) == -1
    assert lib.effort_last_error(None) == b"null context"


def test_shipped_library_reads_no_environment(hip_lib_built):
    """The A/B switches of the lab builds (-DEFFORT_LAB: EFFORT_ABLATE, EFFORT_NO_CUTJOBS ...) are compiled out of the shipped
    library: no EFFORT_* variable name is left in it, and no object of the call path imports getenv (the one import left
    is rocPRIM's own ROCPRIM_USE_ATOMIC_BLOCK_ID inside the offline Q4 converter's radix sort, convert_q4.o)."""
    import subprocess
    blob = open(hip_lib_built, "rb").read()
    for knob in (b"EFFORT_ABLATE", b"EFFORT_NO_CUTJOBS", b"EFFORT_NO_COMPACT_MEANS", b"EFFORT_TAIL_CALLS", b"EFFORT_TAIL_MULT"):
        assert knob not in blob, knob
    csrc = os.path.join(ROOT, "effort_amd", "csrc")
    for obj in ("api.o", "bucket_mul.o", "cutoff.o", "dispatch.o", "decode.o", "gemv.o", "convert.o", "comm.o"):
        path = os.path.join(csrc, obj)
        if os.path.exists(path):
            syms = subprocess.run(["nm", "--undefined-only", path], capture_output=True, text=True, check=True).stdout
            assert "getenv" not in syms, obj


def _multiply_kernel_disassembly(lib_path, tmp_path):
    """Disassembly of the code object of `lib_path` that holds the bucket_mul_kernel instantiations (llvm-objdump --offloading)."""
    import glob
    import shutil
    import subprocess
    llvm = "/opt/rocm/lib/llvm/bin"
    if not os.path.exists(os.path.join(llvm, "llvm-objdump")):
        pytest.skip("no llvm-objdump in this image")
    work = os.path.join(str(tmp_path), os.path.basename(lib_path))
    shutil.copy(lib_path, work)
    subprocess.run([os.path.join(llvm, "llvm-objdump"), "--offloading", work], capture_output=True, cwd=str(tmp_path), check=True)
    for co in sorted(glob.glob(work + ".*gfx950")):
        syms = subprocess.run([os.path.join(llvm, "llvm-readelf"), "-sW", co], capture_output=True, text=True).stdout
        if "bucket_mul_kernel" in syms:
            return subprocess.run([os.path.join(llvm, "llvm-objdump"), "-d", co], capture_output=True, text=True, check=True).stdout
    raise AssertionError("no code object with bucket_mul_kernel in " + lib_path)


def test_shipped_kernels_carry_no_lab_code(hip_lib_built, tmp_path):
    """Round 5's review: the persistent headline instantiation kept runtime tests of the stamp / ablation / trace switches, and the lab
    macros sat beside product code.  Since round 6 the shipped library is the product alone -- its multiply kernels read no clock
    (`s_memrealtime`: the device-clock stamps) and no XCC / HW id (the per-item trace) anywhere in their code -- and the same sources
    built -DEFFORT_LAB are libeffort_hip_lab.so, which does and which the tools load (EFFORT_HIP_LIB=lab)."""
    import ctypes
    import effort_amd._lib as L
    prod = _multiply_kernel_disassembly(hip_lib_built, tmp_path)
    assert prod.count("s_memrealtime") == 0 and "HW_REG_XCC_ID" not in prod and "HW_REG_HW_ID" not in prod
    assert os.path.exists(L.LAB_LIB_PATH), "make all builds the lab library next to the product"
    lab = _multiply_kernel_disassembly(L.LAB_LIB_PATH, tmp_path)
    assert lab.count("s_memrealtime") > 100 and "HW_REG_XCC_ID" in lab
    assert ctypes.CDLL(hip_lib_built).effort_is_lab_build() == 0
    blob = open(L.LAB_LIB_PATH, "rb").read()
    assert b"EFFORT_ABLATE" in blob                                  # (the environment knobs live there, and only there)


def test_shipped_kernels_stream_the_bucket_rows_non_temporally(hip_lib_built, tmp_path):
    """Round 6: the bucket-row stream is read with `nt` (bucket_mul.hip, EFFORT_ROW_AUX: a kept row is read once per call; one 32-call
    launch 160 -> 150 us) -- in every multiply kernel of the shipped library; and since effort_set_row_reuse the kernels that
    serve group launches (the persistent instantiations and the E = 4 plain ones) hold a SECOND copy of the streaming loop with the ordinary
    policy, chosen once per item (a caller whose launches in flight read the same matrices).  The 8-byte (E = 4) row pieces are the only buffer loads of that width the kernels issue besides LDS-direct staging
    (which carries `lds`): half of them carry nt, half do not, and the counted waits of the software pipeline (vmcnt 15 .. 8) survive
    in both copies -- the first form of the switch, a branch per batch, lost them."""
    import re
    prod = _multiply_kernel_disassembly(hip_lib_built, tmp_path)
    body, waits, name = {}, {}, None
    for line in prod.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            name = m.group(1)
        elif name and "bucket_mul_kernel" in name:
            if "buffer_load_dwordx2" in line:
                body.setdefault(name, []).append(line)
            w = re.search(r"s_waitcnt vmcnt\((\d+)\)", line)
            if w:
                waits.setdefault(name, []).append(int(w.group(1)))
    assert len(body) >= 10, "no 8-byte row loads found in the multiply kernels?"
    switched = 0
    for k, lines in body.items():
        m = re.search(r"bucket_mul_kernelILi(\d)ELi(\d+)ELi(\d+)ELb(\d)ELb(\d)ELb(\d)E", k)
        assert m, k
        elems, persist = int(m.group(2)), m.group(6) == "1"
        nt = sum(" nt" in l for l in lines)
        if persist or elems == 4:                       # the kernels of group launches: both policies, chosen per item
            assert nt > 0 and 2 * nt == len(lines), (k, nt, len(lines))
            switched += 1
        else:                                           # the lean kernels a lone call runs: nt only (two copies cost the default path 0.1-0.3 us per call)
            assert nt == len(lines), (k, nt, len(lines))
        assert waits[k].count(15) >= 1 and waits[k].count(12) >= 1, (k, sorted(set(waits[k])))
    assert switched >= 10, switched


def test_shipped_library_has_no_measured_dead_ends(hip_lib_built):
    """What round 4 built and measured SLOWER -- chain launches, the named reducer, byte-indexed Q4 accumulators -- lives on branch
    `chain-launch`, not in the drop-in header or the shipped library; and the library loads without RCCL (multi-GPU binds it at run
    time: a single-GPU host needs no librccl) and without gcc-built extras."""
    import subprocess
    fns = _header_functions()
    assert not [f for f in fns if "chain" in f or "byte_acc" in f], fns
    syms = subprocess.run(["nm", "-D", "--defined-only", hip_lib_built], capture_output=True, text=True, check=True).stdout
    assert "effort_bucketmul_chain" not in syms and "effort_set_q4_byte_acc" not in syms
    kernels = subprocess.run(["nm", "-C", hip_lib_built], capture_output=True, text=True, check=True).stdout
    inst = [l for l in kernels.splitlines() if "bucket_mul_kernel<" in l]
    assert inst, "no multiply kernel in the library?"
    assert all(l.count(",") <= 6 for l in inst), "a CHAIN instantiation (seventh template argument) is back"      # <FMT, E, W, FUSED, COMPACT, PERSIST>
    needed = subprocess.run(["readelf", "-d", hip_lib_built], capture_output=True, text=True, check=True).stdout
    assert "librccl" not in needed, "libeffort_hip.so must not hard-link RCCL (it is opened by effort_comm_*)"
    und = subprocess.run(["nm", "-D", "--undefined-only", hip_lib_built], capture_output=True, text=True, check=True).stdout
    assert "nccl" not in und


def test_product_has_no_oracle_dependency():
    """The product package must never import or link the oracle (it is test infrastructure)."""
    pkg = os.path.join(ROOT, "effort_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.lower(), (dirpath, f)


def test_product_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import effort_amd
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        effort_amd.gpu()


def test_bench_imports_and_prices_the_headline_launch():
    """bench.py must at least import on a machine without a GPU (the driver runs it unattended), and its byte model is the
    SURVEY 8d formula: 4096 x 11008 at 16 737 kept rows = 23 623 008 bytes per call."""
    import importlib
    bench = importlib.import_module("bench")
    assert bench.algorithmic_bytes(16737, 4096, 11008) == 16737 * 688 * 2 + 524288 + 8192 + 16384 + 44032 == 23623008
    assert bench.moved_bytes(16737, 4096, 11008) < bench.algorithmic_bytes(16737, 4096, 11008)
    for fn in ("timeit_protocol", "measured_traffic", "cpu_baseline", "oracle_outputs", "main"):
        assert callable(getattr(bench, fn))
    assert bench.LDS_ATOMIC_PEAK > 1000


def test_bench_compact_line_fits_the_driver_tail():
    """The driver keeps about 8 KB of stdout tail and parses the LAST line (round 4's single 23 KB line left BENCH_r04.parsed =
    null).  bench.compact_line() must turn a full record -- round 4's own, and one bloated far beyond it -- into a line below 4 KB
    that round-trips through json.loads and still carries the contract keys, `roofline` and `cpu_baseline`."""
    import importlib
    import json
    import os
    bench = importlib.import_module("bench")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "profiles", "r04_bench_driver_style.json")) as f:
        full = json.load(f)
    bloated = json.loads(json.dumps(full))
    bloated["config"]["workload"] = "x" * 5000
    bloated["roofline"]["traffic_source"] = "y" * 5000
    bloated["cpu_baseline"]["sample"] = "z" * 5000
    bloated["multi_gpu"] = {"ms_per_step_kernel_only": 1.0, "ms_per_step_with_all_gather": 1.1, "per_shape": {"a": "b" * 9000},
                            "columns": {"ms_per_step_kernel_only": 0.5, "partition": "c" * 3000},
                            "layer_latency": {"effort": 0.5, "us_per_layer_kernel_only": 80.0, "us_per_layer_with_gathers": 120.0, "note": "n" * 3000}}
    for rec in (full, bloated, {"metric": "m", "value": 1.0}):
        line = bench.compact_line(rec)
        assert "\n" not in line and len(line) < bench.COMPACT_LIMIT <= 4096
        got = json.loads(line)
        for k in ("metric", "value"):
            assert got[k] == rec[k]
    got = json.loads(bench.compact_line(full))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in got, k
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_us", "bytes_per_launch"):
        assert k in got["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in got["cpu_baseline"], k
    assert abs(got["roofline"]["frac"] - got["roofline"]["achieved"] / got["roofline"]["peak"]) < 1e-3
    assert "workload" in got["config"] and "model" not in got["config"]


def test_bench_guard_relays_the_record_or_falls_back(monkeypatch, capsys):
    """bench.py at N = 1 measures in a child process: a child that dies without a record (an auxiliary section taking the process
    down) is followed by the contract-complete subset, whose line is relayed with a note naming what happened."""
    import json
    import subprocess
    import sys
    import types
    sys.path.insert(0, ROOT)
    import bench
    calls = []
    record = {"metric": "m", "value": 1.0, "unit": "GB/s", "roofline": {"frac": 0.5}, "cpu_baseline": {"value": 1.0}}

    def fake_run(cmd, env=None, stdout=None):
        calls.append(cmd)
        if len(calls) == 1:
            return types.SimpleNamespace(returncode=-6, stdout=b"RCCL banner\n")
        return types.SimpleNamespace(returncode=0, stdout=("noise\n" + json.dumps(record) + "\n").encode())
    monkeypatch.setattr(subprocess, "run", fake_run)
    assert bench.guarded() == 0
    out = capsys.readouterr().out.strip().split("\n")
    assert len(out) == 1
    line = json.loads(out[0])
    assert line["value"] == 1.0 and "exit code -6" in line["note"]
    assert "--no-sweep" in calls[1] and "--no-sweep" not in calls[0]


def _run_bench(args, env_extra, timeout=180):
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "BENCH_CHILD", "BENCH_NO_GUARD")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` with no launcher around it starts its N ranks itself (one process each, RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* set) and relays rank 0's ONE line with n_gpus = N.  Run here over gloo with the stub step
    (BENCH_STUB=1: the launch contract -- rendezvous, warm-up, barrier, K steps, barrier, max over ranks -- without a GPU)."""
    import json
    p = _run_bench(["--gpus", "2", "--steps", "5", "--warmup", "2"], {"BENCH_STUB": "1"})
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.strip().split("\n") if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["steps"] == 5 and d["warmup"] == 2
    assert d["config"]["partition"] == "matrices" and "launch_ranks" in d["launched_by"] and d["value"] > 0
    for k in ("metric", "value", "unit", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k


def test_bench_honours_an_external_launcher():
    """Under torch.distributed.run (the driver's N > 1 command) the launcher's environment is used as it is: no second level of ranks."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "BENCH_CHILD")}
    env["BENCH_STUB"] = "1"
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29533",
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1"], env=env, capture_output=True, text=True, timeout=240)
    assert p.returncode == 0, p.stderr[-2000:]
    recs = [json.loads(ln) for ln in p.stdout.strip().split("\n") if ln.startswith("{")]
    assert len(recs) == 1 and recs[0]["n_gpus"] == 2 and "launched_by" not in recs[0]


def test_bench_launcher_reports_failures_in_one_line():
    """Fewer GPUs than asked for, or a rank that dies: ONE contract-shaped JSON line with `error` and a null value, never a hang."""
    import json

    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("a multi-GPU box would really launch")
    p = _run_bench(["--gpus", "2"], {})
    d = json.loads(p.stdout.strip().split("\n")[-1])
    assert p.returncode == 2 and d["value"] is None and d["n_gpus"] == 2 and "GPU(s) visible" in d["error"]
    p = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1"], {"BENCH_STUB": "1", "BENCH_STUB_FAIL_RANK": "1", "BENCH_LAUNCH_TIMEOUT_S": "60"})
    d = json.loads(p.stdout.strip().split("\n")[-1])
    assert p.returncode == 1 and d["value"] is None and "rank 1 exited with code 3" in d["error"]
