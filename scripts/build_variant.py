"""Builds an experimental copy of the library with extra compiler flags (the product sources are copied into a scratch
directory, never patched in place):  python scripts/build_variant.py NAME -DFLAG1 -DFLAG2 ...
-> sleqp_amd/_exp_NAME/libhipfact.so; run anything with HIPFACT_LIBRARY=<that path>."""
import os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name, flags = sys.argv[1], sys.argv[2:]
dst = os.path.join(ROOT, "sleqp_amd", "_exp_" + name)  # sibling of csrc: same relative include paths
shutil.rmtree(dst, ignore_errors=True)
os.makedirs(os.path.dirname(dst), exist_ok=True)
shutil.copytree(os.path.join(ROOT, "sleqp_amd", "csrc"), dst, ignore=shutil.ignore_patterns("*.so", "*.o"))
subprocess.check_call(["make", "-C", dst, "-j4", "CXXFLAGS=-O3 -std=c++17 -fPIC -pthread " + " ".join(flags)])
print("built", os.path.join(dst, "libhipfact.so"))
