"""A/B of the shadow rows (ALTRO_NO_SHADOW) on the headline shape, alternating runs in separate processes: kernel ms of the
fused launch at 20 and 100 steps."""
import sys, os, subprocess
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ab = os.path.join(R, "tools", "gpu_ab.py")
for S in (20, 100):
    for rep in range(3):
        for tag, env in (("shadow rows", {}), ("no shadow rows", {"ALTRO_NO_SHADOW": "1"})):
            e = dict(os.environ); e.update(env)
            subprocess.run([sys.executable, ab, "12", "4", "50", "8192", str(S), "%s, %d steps" % (tag, S)], env=e)
