"""python tools/exp/run_patched.py <script.py> [args ...] — the script (bench.py, a tool) with tools/exp/rowmajor_v_patch.py installed first."""
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import rowmajor_v_patch  # noqa: E402

rowmajor_v_patch.install()
sys.argv = sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
