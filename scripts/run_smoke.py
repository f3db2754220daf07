"""python scripts/run_smoke.py: __graft_entry__.smoke() from a GPU job (scripts/gpu_job.sh "py smoke scripts/run_smoke.py")"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

g.smoke()
print("smoke ok")
