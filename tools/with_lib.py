"""A/B helper: run a repo script against an alternative build of libpcdm.so.

    python tools/with_lib.py pcdms_amd/lib/libpcdm_old.so bench.py --no-cpu-baseline --no-vae
"""
import runpy
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pcdms_amd import _lib  # noqa: E402

_lib.load(sys.argv[1])
script = sys.argv[2]
sys.argv = sys.argv[2:]
runpy.run_path(script, run_name="__main__")
