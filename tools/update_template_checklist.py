"""Regenerates tests/golden/bench_kernel_templates.json from the dumps a GPU run of the two checklist tests wrote:
  YV1_DUMP_TEMPLATES=gpurun_out/x/templates.json python -m pytest tests/test_gpu_bench_configs.py tests/test_gpu_bench_configs_fp8.py -m gpu
  python tools/update_template_checklist.py gpurun_out/x/templates.json
Rule: weight-gradient strings of the "[shared entry]" (yv1_conv2d_wgrad_shared_nhwc_bf16, what the training step dispatches on its
side stream) and k_wgrad_stem are bench-dispatched, the stand-alone entry's are other_known; convolution templates keep the class
they had (new ones are listed as other_known until a committed rocprof summary of the bench names them -- move them by hand);
fp8: templates seen by the training form (fwd+stats) are bench-dispatched."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = os.path.join(ROOT, "tests", "golden", "bench_kernel_templates.json")
old = json.load(open(path))
seen = json.load(open(sys.argv[1]))
bench, other = [], []
for k, kinds in sorted(seen.items()):
    if k.startswith("k_conv"):
        (bench if (k in old["bench_dispatched"] or "--all-bench" in sys.argv) else other).append(k)
    elif k.endswith("[shared entry]") or k == "k_wgrad_stem":
        bench.append(k)
    else:
        other.append(k)
old["bench_dispatched"], old["other_known"] = bench, other
f8 = sys.argv[1] + ".fp8"
if os.path.exists(f8):
    s8 = json.load(open(f8))
    old["fp8_bench_dispatched"] = sorted(k for k, kinds in s8.items() if "fwd+stats" in kinds)
    old["fp8_other_known"] = sorted(k for k, kinds in s8.items() if "fwd+stats" not in kinds)
json.dump(old, open(path, "w"), indent=1)
print("bench_dispatched %d, other_known %d, fp8 %d + %d" % (len(bench), len(other), len(old["fp8_bench_dispatched"]), len(old["fp8_other_known"])))
