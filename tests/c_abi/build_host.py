"""Builds tests/c_abi/abi_host.c -- the torch-free C caller of libanyloc_hip.so -- with gcc:
include/anyloc_hip.h + the HIP runtime's C API on one side, the C restatement of the reference (oracle/c, the checker)
on the other.  ``python tests/c_abi/build_host.py`` or ``build_host()``; __graft_entry__.build() calls it so the binary
travels to the GPU box with the snapshot."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "build")
BIN = os.path.join(OUT, "abi_host")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")


def build_host(force=False, verbose=False):
    src = [os.path.join(HERE, "abi_host.c"), os.path.join(ROOT, "oracle", "c", "anyloc_oracle.c")]
    # (the library itself is not a rebuild trigger: the program checks anyloc_version() against the header at start-up)
    deps = src + [os.path.join(ROOT, "include", "anyloc_hip.h"), os.path.join(ROOT, "oracle", "c", "anyloc_oracle.h")]
    os.makedirs(OUT, exist_ok=True)
    if force or not os.path.exists(BIN) or any(os.path.getmtime(d) > os.path.getmtime(BIN) for d in deps):
        cmd = [os.environ.get("CC", "gcc"), "-O2", "-std=c11", "-Wall", "-Wextra", "-fopenmp", "-ffp-contract=off",
               "-D__HIP_PLATFORM_AMD__",            # the HIP headers' own platform selector; this repo has no other platform
               f"-I{ROCM}/include", f"-I{ROOT}/include", f"-I{ROOT}/oracle/c"] + src + \
              ["-o", BIN, f"-L{ROOT}/anyloc_amd", "-lanyloc_hip", f"-L{ROCM}/lib", "-lamdhip64", "-lm",
               "-Wl,-rpath,$ORIGIN/../../../anyloc_amd", f"-Wl,-rpath,{ROCM}/lib"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return BIN


if __name__ == "__main__":
    print(build_host(force=True, verbose=True))
