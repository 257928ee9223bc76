"""The oracle is test infrastructure: nothing in the product package may import, call or execute it,
and the product must not carry a CPU fallback that would let parity claims route around the HIP path."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _py_files(top):
    for d, _, fs in os.walk(top):
        if "__pycache__" in d:
            continue
        for f in fs:
            if f.endswith(".py"):
                yield os.path.join(d, f)


def test_product_package_never_touches_the_oracle():
    offenders = []
    for path in _py_files(os.path.join(ROOT, "mmf_amd")):
        src = open(path).read()
        if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M) or "visual_bert_oracle" in src:
            offenders.append(os.path.relpath(path, ROOT))
    assert not offenders, offenders


def test_only_tests_smoke_and_bench_baseline_import_the_oracle():
    allowed_prefixes = ("tests" + os.sep, "oracle" + os.sep)
    allowed_files = {"bench.py", "__graft_entry__.py"}
    offenders = []
    for path in _py_files(ROOT):
        rel = os.path.relpath(path, ROOT)
        if rel.startswith(allowed_prefixes) or rel in allowed_files or rel.startswith("gpurun_out"):
            continue
        if re.search(r"^\s*(from|import)\s+oracle\b", open(path).read(), flags=re.M):
            offenders.append(rel)
    assert not offenders, offenders
    bench = open(os.path.join(ROOT, "bench.py")).read()
    # in bench.py the oracle appears only inside cpu_baseline()
    body = bench[bench.index("def cpu_baseline"):bench.index("def main")]
    assert bench.count("from oracle") == body.count("from oracle") == 1


def test_product_has_no_eager_math_fallback():
    """functional.py may use torch for allocation / views / autograd only: no matmul, softmax, layer_norm, gelu ..."""
    src = open(os.path.join(ROOT, "mmf_amd", "functional.py")).read()
    for banned in ("torch.matmul", "F.linear", "torch.softmax", "F.softmax", "layer_norm(", "F.gelu", "torch.nn.functional",
                   "torch.bmm", "torch.mm(", "F.dropout", "F.embedding"):
        assert banned not in src, banned
