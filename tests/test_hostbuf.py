"""csrc/hostbuf.hpp on the CPU: the allocation-failure-as-a-value containers that replaced std::vector / std::unordered_map behind the
C ABI (round 5).  Header-only host C++: built with g++, no GPU."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_hostbuf_containers(tmp_path):
    exe = str(tmp_path / "hostbuf_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-fno-exceptions", os.path.join(ROOT, "tests", "cabi", "hostbuf_check.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "hostbuf ok" in out.stdout, out.stdout + out.stderr


def test_no_throwing_containers_left_behind_the_abi():
    """The three translation units that implement the C ABI hold no std::vector / std::map / std::string / std::thread any more."""
    import re
    for unit in ("lto_api.hip", "lto_group.hip", "lto_comm.hip"):
        src = open(os.path.join(ROOT, "lowthrustopt_amd", "csrc", unit)).read()
        code = re.sub(r"//[^\n]*", "", src)                       # comments may name what was replaced
        for bad in ("std::vector", "std::unordered_map", "std::map<", "std::string", "std::thread", "#include <vector>"):
            assert bad not in code, (unit, bad)
