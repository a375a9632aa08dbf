import sys, os
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scenes"))
import make_scenes as ms
ms.HERE = sys.argv[1]
for n in (24, 32, 48, 64, 96, 128):
    cm, co = ms.cloud(n)
    ms.emit("cloud%d.txt" % n, cm, dict(ms.SAMPLE_CAMERA, res=(1920, 1080), iterations=16, file="c.bmp"), co)
