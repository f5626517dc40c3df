R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp PYTHONPATH=$R
for g in 1 0; do for ov in 1 0; do PIDM_GRAPH=$g PIDM_NO_OVERLAP=$ov python tools/r03_host_probe.py 64 2>/dev/null | tail -1; done; done
python - <<'PY'
import ctypes, os
for n in ("libamdhip64.so",):
    pass
import torch
print("torch hip:", torch.version.hip)
os.system("cat /proc/self/maps > /dev/null")
import subprocess
print(subprocess.run("python - <<'Q'\nimport torch,os\nimport physicsinformeddiffusionmodels_amd._lib as l\nl.get_lib()\nprint([x.split()[-1] for x in open('/proc/self/maps') if 'amdhip' in x][:3])\nQ", shell=True, capture_output=True, text=True).stdout)
PY
