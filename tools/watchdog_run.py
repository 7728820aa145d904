"""Run a script with a faulthandler watchdog: dumps every thread's Python stack after N seconds and
exits, so a hang on the GPU box leaves a trace instead of a silent timeout.
usage: python tools/watchdog_run.py SECONDS script.py [args...]"""
import faulthandler
import runpy
import sys

secs = float(sys.argv[1])
faulthandler.dump_traceback_later(secs, exit=True)
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name='__main__')
