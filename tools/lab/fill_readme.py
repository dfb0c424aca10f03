"""Fill the @@...@@ placeholders of README.md from the final battery's artefacts (gpurun_out/<tag>/ and gpurun_out/r6check/)."""
import json
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else 'r6final'


def line(path):
    rows = [x for x in open(path) if x.startswith('{')]
    return json.loads(rows[-1])


d = line(f'gpurun_out/{tag}/bench_default.json')
b128 = line(f'gpurun_out/{tag}/bench_b128.json')
r512 = line(f'gpurun_out/{tag}/bench_xl2_512.json')
log = open(f'gpurun_out/{tag}/gputests_final.log').read()
m = re.search(r'(\d+) passed(?:, (\d+) skipped)?', log)
tests = f'{m.group(1)} passed, {m.group(2) or 0} skipped' if m else 'see profiles/r6_gputests_final.log'
s = open('README.md').read()
sm = d['sampler']
rep = {'IMG': f"{d['value']:.0f}", 'MS': f"{d['ms_per_step']:.1f}", 'NT8': f"{d['roofline']['frac']:.3f}",
       'ENC': f"{d['roofline']['encoder']['frac']:.3f}", 'ENCMS': f"{d['roofline']['encoder']['ms']:.1f}",
       'SMP': f"{sm['value']:.2f}", 'SMP32': f"{sm['fp32_value']:.2f}", 'TF32': f"{sm['fp32']['model_tflops_per_s']:.1f}",
       'FR32': f"{sm['fp32']['model_tflops_per_s'] / 157.3:.3f}", 'B128': f"{b128['ms_per_step']:.1f}", 'R512': f"{r512['value']:.1f}",
       'CPU': f"{d['cpu_baseline']['value']:.1f}", 'TESTS': tests}
for k, v in rep.items():
    s = s.replace(f'@@{k}@@', v)
left = re.findall(r'@@\w+@@', s)
assert not left, left
open('README.md', 'w').write(s)
print(rep)
