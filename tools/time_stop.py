"""classify_signal fused (125 000 clips) timing, as bench.py --workload stop does"""
import subprocess, sys, json, os
r = subprocess.run([sys.executable, "bench.py", "--workload", "stop", "--no-cpu-baseline", "--steps", "30"], capture_output=True, text=True)
d = json.loads(r.stdout.strip().splitlines()[-1]); print("stop %.4f ms" % d["roofline"]["kernel_ms"])
