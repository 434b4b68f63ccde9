"""GPU: the job set-ups of the reference's own smoke tests (tests/test_0030_align.py, test_0040_balance.py,
test_0050_align_balance.py, test_0060_stack.py, test_0061_depth_map.py) RE-TYPED against `shinestacker_amd`: the same
public-API calls with the same arguments ("exactly as tests/test_0060_stack.py:7-14"), condensed --
parametrised instead of one function per case, no try / except around the runs -- and pointed at a temporary `examples/`
directory holding the committed crops of the reference's example frames (tests/golden/img_jpg_crop) as 8-bit JPEG and as
16-bit TIFF.  What is asserted is what the reference asserts: the jobs run through, outputs have the expected type.  (Plots
and progress bars -- `plot_summary`, `plot_histograms`, `callbacks='tqdm'` -- are accepted and ignored: out of scope,
SURVEY.md 2.)  The same classes are compared value by value with recordings of the reference in test_gpu_parity.py /
test_align_golden.py / test_gpu_balance.py."""
import os
from unittest.mock import MagicMock

import numpy as np
import pytest

from shinestacker_amd import (AlignFrames, BalanceFrames, CombinedActions, DepthMapStack, FocusStack, FocusStackBunch,
                              PyramidStack, StackJob)
from shinestacker_amd.align import align_images
from shinestacker_amd.defaults import constants
from shinestacker_amd.imageio import read_img, write_img

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "img_jpg_crop")


@pytest.fixture(scope="module", autouse=True)
def examples(tmp_path_factory, hiplib):
    """cwd = a directory that looks like the reference's repository root: examples/input/img-jpg, examples/input/img-tif"""
    hiplib.require_device()
    root = tmp_path_factory.mktemp("refsuite")
    for sub in ("img-jpg", "img-tif"):
        os.makedirs(root / "examples" / "input" / sub)
    for k, n in enumerate(sorted(os.listdir(GOLDEN))):
        img = read_img(os.path.join(GOLDEN, n))
        write_img(str(root / "examples" / "input" / "img-jpg" / f"{k:04d}.jpg"), img)
        write_img(str(root / "examples" / "input" / "img-tif" / f"{k:04d}.tif"), img.astype(np.uint16) * 257)
    old = os.getcwd()
    os.chdir(root)
    yield root
    os.chdir(old)


# ---------------------------------------------------------------- tests/test_0060_stack.py
def test_jpg():
    job = StackJob("job", "examples", input_path="input/img-jpg")
    job.add_action(FocusStack("stack-pyramid", PyramidStack(), output_path="output/img-jpg-stack", prefix='pyr_'))
    job.run()
    out = os.listdir("examples/output/img-jpg-stack")
    assert len(out) == 1 and out[0].startswith("pyr_") and read_img(os.path.join("examples/output/img-jpg-stack", out[0])).dtype == np.uint8


def test_tif():
    job = StackJob("job", "examples", input_path="input/img-tif")
    job.add_action(FocusStack("stack-pyramid-tiff", PyramidStack(), output_path="output/img-tif-stack", prefix='pyr_'))
    job.run()
    out = os.listdir("examples/output/img-tif-stack")
    assert len(out) == 1 and read_img(os.path.join("examples/output/img-tif-stack", out[0])).dtype == np.uint16


def test_jpg_dm():
    job = StackJob("job", "examples", input_path="input/img-jpg")
    job.add_action(FocusStack("stack-depthmap", DepthMapStack(), output_path="output/img-jpg-stack", prefix='dm_'))
    job.run()
    assert any(f.startswith("dm_") for f in os.listdir("examples/output/img-jpg-stack"))


def test_bunches():
    job = StackJob("job", "examples", input_path="input/img-jpg")
    job.add_action(FocusStackBunch("stack-pyramid-bunch", PyramidStack(), output_path="output/img-jpg-bunches", frames=3))
    job.run()
    assert len(os.listdir("examples/output/img-jpg-bunches")) == 4      # 6 frames, 3 per bunch, overlap 2


# ---------------------------------------------------------------- tests/test_0030_align.py
def test_align():
    img_1, img_2 = [read_img(f"examples/input/img-jpg/000{i}.jpg") for i in (2, 3)]
    n_good_matches, M, img_warp = align_images(img_1, img_2)
    assert img_warp is not None
    assert n_good_matches > 100


def test_align_homo():
    img_1, img_2 = [read_img(f"examples/input/img-jpg/000{i}.jpg") for i in (2, 3)]
    n_good_matches, M, img_warp = align_images(img_1, img_2, alignment_config={'transform': constants.ALIGN_HOMOGRAPHY})
    assert img_warp is not None and np.asarray(M).shape == (3, 3)
    assert n_good_matches > 10


def test_align_rescale():
    img_1, img_2 = [read_img(f"examples/input/img-jpg/000{i}.jpg") for i in (2, 3)]
    n_good_matches, M, img_warp = align_images(img_1, img_2, alignment_config={'subsample': 4})
    assert img_warp is not None
    assert n_good_matches > 10


def test_align_ecc():
    img_1, img_2 = [read_img(f"examples/input/img-jpg/000{i}.jpg") for i in (2, 3)]
    n_good_matches, M, img_warp = align_images(img_1, img_2, alignment_config={'ecc_refinement': True})
    assert img_warp is not None
    assert n_good_matches > 10


def test_align_jpg_job():
    job = StackJob("job", "examples", input_path="input/img-jpg", callbacks='tqdm')
    job.add_action(CombinedActions("align-jpg", [AlignFrames(plot_summary=True)], output_path="output/img-jpg-align"))
    job.run()
    assert len(os.listdir("examples/output/img-jpg-align")) == 6


# ---------------------------------------------------------------- tests/test_0040_balance.py
@pytest.mark.parametrize("src,channel,corr_map,out", [
    ("img-tif", constants.BALANCE_RGB, constants.BALANCE_MATCH_HIST, "img-tif-balance-rgb-match"),
    ("img-jpg", constants.BALANCE_LUMI, constants.BALANCE_LINEAR, "img-jpg-balance-lumi"),
    ("img-tif", constants.BALANCE_LUMI, constants.BALANCE_GAMMA, "img-tif-balance-lumi"),
    ("img-jpg", constants.BALANCE_RGB, constants.BALANCE_LINEAR, "img-jpg-balance-rgb"),
    ("img-jpg", constants.BALANCE_HSV, constants.BALANCE_LINEAR, "img-jpg-balance-sv"),
    ("img-jpg", constants.BALANCE_HLS, constants.BALANCE_GAMMA, "img-jpg-balance-ls"),
])
def test_balance(src, channel, corr_map, out):
    job = StackJob("job", "examples", input_path=f"input/{src}", callbacks='tqdm')
    job.add_action(CombinedActions("balance", [BalanceFrames(channel=channel, corr_map=corr_map, plot_histograms=True,
                                                             plot_summary=True)], output_path=f"output/{out}"))
    job.run()
    assert len(os.listdir(f"examples/output/{out}")) == 6


# ---------------------------------------------------------------- tests/test_0050_align_balance.py
@pytest.mark.parametrize("kw,out", [
    (dict(channel=constants.BALANCE_HLS, corr_map=constants.BALANCE_GAMMA), "img-jpg-align-balance-ls"),
    (dict(channel=constants.BALANCE_HSV), "img-jpg-align-balance-sv"),
    (dict(channel=constants.BALANCE_RGB), "img-jpg-align-balance-rgb"),
    (dict(channel=constants.BALANCE_LUMI), "img-jpg-align-balance-lumi"),
])
def test_align_balance(kw, out):
    job = StackJob("job", "examples", input_path="input/img-jpg")
    job.add_action(CombinedActions("align", [AlignFrames(), BalanceFrames(**kw)], output_path=f"output/{out}"))
    job.run()
    assert len(os.listdir(f"examples/output/{out}")) == 6


# ---------------------------------------------------------------- tests/test_0061_depth_map.py
def test_initialization():
    dms = DepthMapStack()
    assert dms.map_type == constants.DEFAULT_DM_MAP
    assert dms.energy == constants.DEFAULT_DM_ENERGY


def test_sobel_map_with_examples():
    filenames = [os.path.join("examples/input/img-jpg/", f"000{i}.jpg") for i in range(6)]
    dms = DepthMapStack()
    gray_images = []
    for img_path in filenames[:3]:
        img = read_img(img_path)          # (the reference reads with cv2.IMREAD_GRAYSCALE; any gray plane serves)
        gray_images.append(img[..., 1].astype(np.float32))
    gray_images = np.array(gray_images)
    sobel_map = dms.get_sobel_map(gray_images)
    assert sobel_map.shape == gray_images.shape
    assert sobel_map.dtype == np.float32
    assert np.all(sobel_map >= 0)  # Energy should always be positive


def test_focus_stack_with_examples():
    filenames = [os.path.join("examples/input/img-jpg/", f"000{i}.jpg") for i in range(6)]
    dms = DepthMapStack()
    dms.process = MagicMock()
    dms.process.callback.return_value = True  # Keep running
    dms.print_message = MagicMock()
    result = dms.focus_stack(filenames[:3])
    assert len(result.shape) == 3
    assert result.dtype == np.uint8
    assert not np.array_equal(result, read_img(filenames[0]))
    result = dms.focus_stack(filenames)
    assert result.shape[0] > 0 and result.shape[1] > 0
