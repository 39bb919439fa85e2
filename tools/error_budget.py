"""Launcher of the device error budget (tests/error_budget.py: the shipped arithmetic against the fp64 oracle over >= 16k samples
per variant, weight seed and encoder pin).  The measurement itself is test infrastructure and lives under tests/; this script
only starts it as a child process, before anything here has touched the GPU.

    python tools/error_budget.py --samples 16384 --out gpurun_out/r04_error_budget.json"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

if __name__ == '__main__':
    sys.exit(subprocess.call([sys.executable, '-m', 'tests.error_budget'] + sys.argv[1:], cwd=ROOT))
