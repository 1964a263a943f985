"""Generates tests/golden/cornell_flat.npz: the flat scene arrays of
the reference's only shipped scene, produced by the numpy restatement of the scene ingest
(oracle/gltf_flatten.py) from /root/reference/media/scenes/cornell.gltf + cornell.bin.
The reference tree does not exist on the GPU box, so the derived arrays travel as a fixture.
Run here (reference mounted):  python tests/golden/make_cornell_fixture.py"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gltf_flatten  # noqa: E402

src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/media/scenes/cornell.gltf"
sc = gltf_flatten.load_gltf(src)
sc.save_npz(os.path.join(HERE, "cornell_flat.npz"))
print("wrote cornell_flat.npz:", sc.positions.shape[0], "vertices,", sc.instanced_triangle_count, "instanced triangles")
