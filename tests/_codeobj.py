"""Read the gfx950 code objects embedded in libomg_hip.so (test helper, no GPU needed).

hipcc stores one clang offload bundle per translation unit in the `.hip_fatbin` section; every bundle entry for gfx950 is an AMDGPU ELF
whose NT_AMDGPU_METADATA note lists, per kernel, the register counts, spill counts, LDS and scratch sizes the loader will use.
`kernels(path)` returns {demangled-ish symbol: metadata dict} for all of them."""
import os
import re
import struct
import subprocess
import tempfile

import yaml

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def available(path: str) -> bool:
    return os.path.exists(path) and os.path.exists(READELF)


def _code_objects(path: str):
    sec = subprocess.run([READELF, "-S", "-W", path], capture_output=True, text=True, check=True).stdout
    m = re.search(r"\.hip_fatbin\s+\S+\s+([0-9a-f]+)\s+([0-9a-f]+)\s+([0-9a-f]+)", sec)
    if m is None:
        raise RuntimeError("no .hip_fatbin section in " + path)
    off, size = int(m.group(2), 16), int(m.group(3), 16)
    with open(path, "rb") as f:
        f.seek(off)
        data = f.read(size)
    pos = 0
    while True:
        i = data.find(MAGIC, pos)
        if i < 0:
            return
        n = struct.unpack_from("<Q", data, i + len(MAGIC))[0]
        p, end = i + len(MAGIC) + 8, i + len(MAGIC) + 8
        for _ in range(n):
            o, s, ts = struct.unpack_from("<QQQ", data, p)
            p += 24
            triple = data[p:p + ts].decode()
            p += ts
            if "gfx950" in triple and s:
                yield data[i + o:i + o + s]
            end = max(end, i + o + s)
        pos = end


def kernels(path: str) -> dict:
    out = {}
    with tempfile.TemporaryDirectory() as d:
        for k, blob in enumerate(_code_objects(path)):
            fn = os.path.join(d, f"{k}.co")
            with open(fn, "wb") as f:
                f.write(blob)
            notes = subprocess.run([READELF, "--notes", fn], capture_output=True, text=True, check=True).stdout
            m = re.search(r"^\s*---\n(.*?)^\.\.\.", notes, re.S | re.M)
            if m is None:
                continue
            for kd in yaml.safe_load(m.group(1)).get("amdhsa.kernels", []):
                out[kd[".name"]] = {key.lstrip("."): val for key, val in kd.items() if key != ".args"}
    return out


OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def disassembly(path: str, needle: str, operands: bool = False) -> dict:
    """{kernel symbol: [mnemonic, ...]} for every kernel whose name contains ``needle`` (llvm-objdump -d of the embedded objects);
    ``operands=True`` keeps the whole instruction text ("v_mfma_f32_32x32x16_f16 a[0:15], v[2:5], ...") instead of the mnemonic."""
    out = {}
    with tempfile.TemporaryDirectory() as d:
        for k, blob in enumerate(_code_objects(path)):
            if needle.encode() not in blob:
                continue
            fn = os.path.join(d, f"{k}.co")
            with open(fn, "wb") as f:
                f.write(blob)
            txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", fn], capture_output=True, text=True, check=True).stdout
            cur = None
            for line in txt.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
                if m:
                    cur = m.group(1) if needle in m.group(1) else None
                    if cur is not None:
                        out[cur] = []
                elif cur is not None:
                    t = line.strip()
                    if t and not t.startswith(("//", ";")):
                        out[cur].append(t.split("//")[0].strip() if operands else t.split()[0])
    return out
