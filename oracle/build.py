"""Builds the checker (oracle/liboracle.so, and oracle/_ref/ where a checkout of the reference is present) for the CPU it runs on.
Test infrastructure: imported by tests/conftest.py, bench.py's cpu_baseline leg and __graft_entry__ only.

The Makefile compiles with -march=native, so a library built on another machine (the build container's, travelling to the GPU box with
the tree) is rebuilt where it runs: liboracle.so.cpu names the CPU the library was compiled on and for."""
import hashlib
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "liboracle.so")


def cpu_stamp():
    model, flags = "", ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and not model:
                model = line.split(":", 1)[1].strip()
            elif line.startswith("flags") and not flags:
                flags = line.split(":", 1)[1].strip()
            if model and flags:
                break
    except OSError:
        pass
    return f"{model} {hashlib.sha1(flags.encode()).hexdigest()[:12]}"


def ensure_built(force=False):
    srcs = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith((".cpp", ".hpp"))]
    srcs.append(os.path.join(os.path.dirname(HERE), "include", "zygpu.h"))
    stamp = cpu_stamp()
    try:
        built_for = open(SO + ".cpu").read()
    except OSError:
        built_for = None
    stale = force or not os.path.exists(SO) or built_for != stamp or any(os.path.getmtime(s) > os.path.getmtime(SO) for s in srcs)
    if stale:
        subprocess.run(["make", "-C", HERE, "-B", "liboracle.so"], check=True, capture_output=True)
        open(SO + ".cpu", "w").write(stamp)
    subprocess.run(["make", "-C", HERE, "ref"], check=True, capture_output=True)
    return SO
