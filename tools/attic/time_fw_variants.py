"""Runs tools/time_fw.py once per library variant in tools/_v/dp_*.so (children, selected with VLGAE_AMD_LIB)."""
import glob, os, subprocess, sys
for so in sorted(glob.glob('tools/_v/dp_*.so')):
    print(os.path.basename(so), end=': ', flush=True)
    subprocess.run([sys.executable, 'tools/time_fw.py'], env=dict(os.environ, VLGAE_AMD_LIB=os.path.abspath(so)))
