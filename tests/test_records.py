"""CPU: the committed evidence matches the committed sources.  bench.py refuses PMC traffic figures measured on other kernel
sources (roofline.traffic = null + the reason); round 5 shipped exactly that -- a kernel edit after the last profiling run.
This test fails in that state, so it cannot reach a round's end unnoticed: re-run tools/profile_r06.sh on the GPU box, then
tools/promote_r06.sh here."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_traffic_json_was_measured_on_these_kernel_sources():
    import bench
    with open(os.path.join(ROOT, "profiles", "traffic.json")) as fh:
        tj = json.load(fh)
    sha = bench.kernel_source_sha()
    for arith in ("separable", "exact"):
        ent = tj[arith]
        assert ent["source_sha"] == sha, (f"profiles/traffic.json[{arith}] was measured on kernel sources {ent['source_sha']}, the "
                                           f"tree holds {sha}: tools/profile_r06.sh (GPU box), then tools/promote_r06.sh")
        assert ent["dtype"] == "f32" and ent["frames_per_launch"] == 16 and ent["hbm_bytes_per_launch"] > 1e9
    # the separable entry is the kernel the default bench launches for level 0 (levels 0 / 1 as a pair for float-32 batches)
    assert "level_sep_pair<float, true" in tj["separable"]["kernel"]


def test_round_profiles_are_tracked():
    """the rocprofv3 summaries the bench line's figures come from are in the tracked tree (profiles/, not gpurun_out/)"""
    d = os.path.join(ROOT, "profiles", "r06")
    have = set(os.listdir(d))
    for f in ("rocprofv3_summary.txt", "rocprofv3_summary_nopair.txt", "rocprofv3_summary_u8.txt", "rocprofv3_summary_u16.txt",
              "rocprofv3_summary_exact.txt", "kernel_stats.csv", "timeline.txt", "bench_default.json"):
        assert f in have, f
    assert any(f.startswith("pair_ab_") for f in have)


def test_cpu_baseline_leg_runs_for_every_input_type():
    """bench.py's cpu_baseline on a tiny sample (CPU only): the streaming port on the GPU leg's own frame type (float-32 frames are
    level 0 as they are), the reference-shaped leg on the integer frames the reference reads -- the default line must not
    die in its side measurements"""
    import types
    import bench
    vals = {}
    for dt in ("f32", "u8", "u16"):
        a = types.SimpleNamespace(cpu_frames=3, height=96, width=160, dtype=dt, arith="separable", cpu_refshaped=2)
        out = bench.cpu_baseline(a, 6)
        assert out["value"] > 0 and out["kind"] == "port" and out["cores"] >= 1
        assert out["reference_shaped"]["value"] > 0 and out["reference_shaped"]["frames"] == 2
        vals[dt] = out
    assert "float32" in vals["f32"]["sample"] and "uint8" in vals["u8"]["sample"] and "uint16" in vals["u16"]["sample"]
