"""Regenerates the prototype section of include/ctclip_hip.h from the extern "C" definitions in ct_clip_amd/csrc/*.hip,
keeping the hand-written per-function comments that are already in the header (new functions get an empty comment to fill in)."""
import glob
import re

HDR = "include/ctclip_hip.h"
old = open(HDR).read()
pre = old[:old.index("typedef struct ihipStream_t* hipStream_t;") + len("typedef struct ihipStream_t* hipStream_t;\n")]
comments = {m.group(2): m.group(1) for m in re.finditer(r"/\* ([^\n]*?) \*/\n[^\n]*?(ctclip_\w+)\(", old)}
protos = []
for f in sorted(glob.glob("ct_clip_amd/csrc/*.hip")):
    s = open(f).read()
    for m in re.finditer(r'extern "C" ([^{;]+?)\s*\{', s):
        p = " ".join(m.group(1).split())
        if "ctclip_set_error" in p:
            continue
        protos.append(p)
out = [pre]
for p in protos:
    name = re.search(r"(ctclip_\w+)\(", p).group(1)
    out.append(f"/* {comments.get(name, 'TODO: document')} */")
    out.append(p.replace("()", "(void)") + ";\n")
out.append("#ifdef __cplusplus\n}\n#endif\n#endif\n")
open(HDR, "w").write("\n".join(out))
print(len(protos), "prototypes")
