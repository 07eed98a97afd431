cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, subprocess, os, pathlib
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
from test_facade_gpu import build_facade_exe
exe = build_facade_exe(pathlib.Path('/tmp'), 'player_controller.cpp')
env = dict(os.environ, PLAYER_DBG=os.environ.get('PLAYER_DBG', '215'))
r = subprocess.run([exe], capture_output=True, text=True, env=env)
print(r.stdout[-6000:], r.stderr[-500:])
PY
