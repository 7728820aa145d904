"""Source stamps for a committed counter survey: `python tools/stamp_sources.py hotformerloc_amd/csrc/attention.hip ... >> profiles/x.txt`
appends one `kernel_source_sha1 <path> <sha1>` line per file; bench.py reports a survey's figure only while the stamped
sources are unchanged (PMC values must not silently go stale in the bench line)."""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for rel in sys.argv[1:]:
    print('kernel_source_sha1', rel, hashlib.sha1(open(os.path.join(ROOT, rel), 'rb').read()).hexdigest())
