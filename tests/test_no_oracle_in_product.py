"""The oracle is test infrastructure: nothing under torchfx_amd/ (Python or HIP) may import,
link or call it, and the product must not read /root/reference at run time."""
import os
import re

from tests.conftest import ROOT


def product_files():
    for d, _, fs in os.walk(os.path.join(ROOT, "torchfx_amd")):
        if "build" in d.split(os.sep):
            continue
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                yield os.path.join(d, f)


def test_product_never_touches_the_oracle_or_the_reference():
    bad = re.compile(r"(^|\W)(import\s+oracle|from\s+oracle|liboracle|oracle/|/root/reference)")
    for p in product_files():
        for i, line in enumerate(open(p, encoding="utf-8", errors="replace"), 1):
            code = line.split("#")[0] if p.endswith(".py") else line
            assert not bad.search(code), f"{p}:{i}: {line.strip()}"


def test_no_cpu_fallback_branches():
    """torchfx_ext must not be able to compute on the host: no SciPy / torch.nn.functional / torch.fft
    imports, every tensor op goes to the compiled extension, and every kernel of the extension checks
    the device before anything else."""
    import ast
    src = open(os.path.join(ROOT, "torchfx_amd", "torchfx_ext.py")).read()
    tree = ast.parse(src)
    mods = set()
    for node in ast.walk(tree):
        if isinstance(node, ast.Import):
            mods |= {a.name for a in node.names}
        elif isinstance(node, ast.ImportFrom):
            mods.add(node.module or "")
    assert not any(m.startswith(("scipy", "torch.nn", "torch.fft", "oracle")) for m in mods), mods
    ops = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name.endswith("_forward")]
    assert len(ops) >= 13                # every device op of the boundary
    for fn in ops:
        assert "native.ops()" in ast.unparse(fn), fn.name
    cpp = open(os.path.join(ROOT, "torchfx_amd", "csrc", "ext", "torchfx_ext.cpp")).read()
    cuda_block = cpp[cpp.index("TORCH_LIBRARY_IMPL(torchfx_hip, CUDA, m)"):cpp.index("TORCH_LIBRARY_IMPL(torchfx_hip, Meta, m)")]
    impls = re.findall(r'm\.impl\("(\w+)", (\w+)\)', cuda_block)
    assert len(impls) >= 14
    for name, fn in impls:
        body = cpp[cpp.index(" " + fn + "("):]
        body = body[:body.index("\n}\n")]
        assert "need_device(" in body or "_impl(" in body or "deinterleave_into_op(" in body, (name, fn)
    assert cpp.count("need_device(") >= 12


def test_reference_build_is_fenced_to_the_cpu_baseline_leg():
    """oracle/_ref/torchfx_ext.so (the reference's own C++ compiled by oracle/Makefile) travels to the GPU box for ONE purpose:
    bench.py's `cpu_baseline.reference_iir_stage`, which runs oracle/ref_time.py in a subprocess.  No `-m gpu` test, shared GPU
    helper or `smoke()` may open it or that script; the only other users are the build-container generators under oracle/."""
    import ast
    import glob
    opens = re.compile(r"oracle/_ref|[\"']_ref[\"']|ref_time|_ref/torchfx_ext")
    for p in sorted(glob.glob(os.path.join(ROOT, "tests", "test_gpu_*.py"))) + [os.path.join(ROOT, "tests", "gpu_common.py")]:
        for i, line in enumerate(open(p, encoding="utf-8"), 1):
            assert not opens.search(line), f"{p}:{i}: {line.strip()}"
    entry = ast.parse(open(os.path.join(ROOT, "__graft_entry__.py")).read())
    smoke = next(n for n in entry.body if isinstance(n, ast.FunctionDef) and n.name == "smoke")
    assert not opens.search(ast.unparse(smoke))
    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    users = {n.name for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and opens.search(ast.unparse(n))}
    assert users == {"cpu_baseline", "run_ref"}, users
    users_oracle = [f for f in glob.glob(os.path.join(ROOT, "oracle", "*.py")) if opens.search(open(f).read())]
    assert {os.path.basename(f) for f in users_oracle} <= {"make_golden.py", "ref_time.py", "check_reference_binding.py"}
