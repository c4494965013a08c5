#!/bin/bash
# usage: tools/gz_ratio.sh   BGZF ratio and kernel times of the C2 text under SG_GZ_* experiment knobs (one line per setting)
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  env $v python3 - "$v" <<'PY'
import json, os, subprocess, sys
r = subprocess.run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--strong-scale", "0", "--no-cpu-baseline", "--no-md5"], capture_output=True, text=True)
d = json.loads(r.stdout.strip().splitlines()[-1])
hp = d["host_pinned"]
print(sys.argv[1], "ratio %.3f" % (hp["plain"]["bytes_per_step"] / hp["gzip"]["bytes_per_step"]), "gzip %.1f M pairs/s" % (hp["gzip"]["value"] / 1e6),
      "ms/step %.2f" % hp["gzip"]["ms_per_step"])
PY
done
