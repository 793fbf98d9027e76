"""Host (no GPU): the compiled ISA of the sliding-window convolution kernels keeps what their design depends on -- every band
loop is ONE basic block with all its MFMAs, and every register a buffer load fills in one band and the next band reads is covered
by an explicit, correctly counted `s_waitcnt vmcnt(N)` (the compiler's own pass does not cover loop-carried loads there, see
tedm_amd/csrc/conv_slide.hip).  tools/check_slide_waits.py compiles both sources with hipcc (cross-compilation) and proves it
from the disassembly."""
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def test_band_loops_are_single_blocks_with_covered_loads():
    r = subprocess.run([sys.executable, str(ROOT / 'tools' / 'check_slide_waits.py')], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith('ok'), r.stdout[-3000:] + r.stderr[-2000:]
    assert r.stdout.count('band loop') == 8 and 'NOT COVERED' not in r.stdout
