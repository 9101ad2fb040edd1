"""Host loader mirror (load.rs) against the facts of the shipped scene."""
import os

import numpy as np
import pytest

from conftest import GOLDEN


def test_teapot_obj(desc):
    m = desc.meshes[0]
    assert m.verts.shape == (3241, 3) and m.normals.shape == (3242, 3) and m.uvw.shape == (1588, 3)
    assert m.tris.shape == m.ntris.shape == m.ttris.shape == (6320, 3)      # 3120 quads * 2 + 80 triangles
    # wavefront_obj 5.1 fan around the LAST vertex: first face "f 1/1/1 2/2/2 3/3/3 4/4/4" -> (3,0,1),(3,1,2)
    assert m.tris[0].tolist() == [3, 0, 1] and m.tris[1].tolist() == [3, 1, 2]
    assert m.tris.max() == 3240 and m.ntris.max() == 3241 and m.ttris.max() == 1587


def test_scene_xml(desc):
    names = [n.name for n in desc.nodes]
    assert names == ["light", "box", "WallBottom", "WallTop", "WallBack", "WallFront", "WallLeft", "WallRight", "teapot.obj", "sphere2"]
    assert [n.parent for n in desc.nodes] == [-1, -1, 1, 1, 1, 1, 1, 1, -1, -1]
    assert [n.geom for n in desc.nodes] == [2, 0, 2, 2, 2, 2, 2, 2, 3, 1]
    mats = {m.name: m for m in desc.materials}
    assert list(mats) == ["wall", "wallRed", "wallBlue", "light", "mtl1", "mtl2"]
    np.testing.assert_array_equal(mats["light"].emission, [20, 20, 20])
    np.testing.assert_array_equal(mats["light"].diffuse.color, [0.5, 0.5, 0.5])           # default grey (load.rs:126)
    np.testing.assert_array_equal(mats["mtl2"].diffuse.color, [0, 0, 0])                  # value="0" * colour
    np.testing.assert_allclose(mats["mtl2"].refraction.color, [0.2, 0.8, 0.2])
    assert abs(mats["mtl2"].refraction_index - 1.52) < 1e-6 and abs(mats["mtl2"].reflection_glossiness - 0.1) < 1e-7
    np.testing.assert_array_equal(mats["mtl2"].reflection.color, [1, 1, 1])
    # light: scale 5 then translate (0,-5,23)
    np.testing.assert_array_equal(desc.nodes[0].m, [5, 0, 0, 0, 5, 0, 0, 0, 5])
    np.testing.assert_array_equal(desc.nodes[0].t, [0, -5, 23])
    # WallTop: scale 64, rotate 180 about x, translate z 12 (document order; translate is NOT rotated)
    np.testing.assert_allclose(desc.nodes[3].m.reshape(3, 3).T, np.diag([64, -64, -64]), atol=1e-4)
    np.testing.assert_array_equal(desc.nodes[3].t, [0, 0, 12])
    # teapot: scale .75, rotate z 30 deg, translate
    c, s = np.cos(np.pi / 6) * 0.75, np.sin(np.pi / 6) * 0.75
    np.testing.assert_allclose(desc.nodes[8].m.reshape(3, 3).T, [[c, -s, 0], [s, c, 0], [0, 0, 0.75]], atol=1e-6)
    np.testing.assert_array_equal(desc.nodes[8].t, np.array([-4.5, 5, 0], np.float32))
    cam = desc.camera
    assert (cam.width, cam.height, cam.fov, cam.focaldist, cam.dof) == (800, 600, 30.0, 1.0, 0.0)
    # load.rs:265: up = (dir x up) x dir, NOT renormalised -> |up| = |right| ~ 0.9964
    assert abs(np.linalg.norm(cam.up) - 0.99643) < 1e-4 and abs(np.linalg.norm(cam.dir) - 1) < 1e-6
    assert desc.background.kind == 0 and not desc.background.color.any() and not desc.environment.color.any()


def test_errors(tmp_path):
    from hpt import scene_io
    p = tmp_path / "s.xml"
    p.write_text("<xml><camera/></xml>")
    with pytest.raises(ValueError, match="no <scene> tag found"):
        scene_io.load_scene(str(p))
    p.write_text('<xml><scene><object type="cube" material="m"/></scene><camera/></xml>')
    with pytest.raises(ValueError, match="unknown object type"):
        scene_io.load_scene(str(p))
    p.write_text('<xml><scene><object type="sphere" material="nope"/></scene><camera><position/><target z="-1"/><up/><fov value="40"/><width value="4"/><height value="4"/></camera></xml>')
    with pytest.raises(ValueError, match="material does not exist"):
        scene_io.load_scene(str(p))
    p.write_text('<xml><scene><material type="blinn" name="m"/></scene></xml>')
    with pytest.raises(ValueError, match="no specular"):
        scene_io.load_scene(str(p))


def test_png_roundtrip(tmp_path):
    from hpt import scene_io
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, size=(9, 13, 4), dtype=np.uint8)
    path = str(tmp_path / "x.png")
    scene_io.write_png_rgba8(path, img)
    back = scene_io.read_png_rgb8(path)
    np.testing.assert_array_equal(back, img[:, :, :3])
