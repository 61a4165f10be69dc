"""loadParamsFromConfigFile (libstereo-odometry.h:551-672) behind the C-ABI: svo_params_load_ini reads the reference's INI keys.
Host-only code of libsvo_hip.so: runs without a GPU."""
import ctypes as C

import pytest

from stereo_vo_amd import hip

SECTIONS = ["RECTIFY", "DETECT", "MATCH", "IF-MATCH", "LEAST_SQUARES", "GUI", "GENERAL"]

INI = """\
; a file as the reference's demo would be given one (demo-main.cpp:120-135)
[RECTIFY]
nOctaves = 3

[DETECT]
detect_method = 1            // FAST + ORB
min_distance=5
initial_FAST_threshold = 17
fast_min_th = 7
fast_max_th = 41
orb_nfeats = 1234
orb_nlevels = 6
minimum_ORB_response = 0.0025
non_maximal_suppression = yes
non_max_supp_method = 1
KLT_win = 4                  // off the path: accepted, ignored
# a comment line
[match]
match_method = 1
max_y_diff = 2.5
enable_robust_1to1_match = false
orb_min_th = 20
orb_max_th = 90
ORB_MAX_DISTANCE = 55
sad_max_ratio = 0.3

[IF-MATCH]
window_height = 12
window_width = 34
filter_fund_matrix = 1

[LEAST_SQUARES]
use_previous_pose_as_initial = TRUE
initial_max_iters = 4
max_iters = 33
min_mod_out_vector = 1e-4
max_incr_cost = 7
residual_threshold = 12.5
bad_tracking_th = 9
use_robust_kernel = 0
kernel_param = 2.25

[GUI]
show_gui = true

[GENERAL]
vo_use_matches_ids = no
vo_out_dir = /tmp/x
"""


def test_every_key_of_the_reference_loader_is_read(tmp_path):
    f = tmp_path / "vo.ini"
    f.write_text(INI)
    p = hip.default_params()
    p.ifm_method = 1
    q = hip.load_params_ini(f, SECTIONS, p)
    assert q is p
    got = {k: getattr(p, k) for k, _ in p._fields_ if not k.startswith("_")}
    want = dict(nOctaves=3, detect_method=1, min_distance=5, initial_FAST_threshold=17, fast_min_th=7, fast_max_th=41, orb_nfeats=1234,
                orb_nlevels=6, minimum_ORB_response=0.0025, non_maximal_suppression=1, nmsMethod=1, match_method=1, max_y_diff=2.5,
                enable_robust_1to1_match=0, orb_min_th=20, orb_max_th=90, orb_max_distance=55.0, ifm_win_h=12, ifm_win_w=34,
                filter_fund_matrix=1, use_previous_pose_as_initial=1, initial_max_iters=4, max_iters=33, min_mod_out_vector=1e-4,
                max_incr_cost=7, residual_threshold=12.5, bad_tracking_th=9, use_robust_kernel=0, kernel_param=2.25, vo_use_matches_ids=0,
                ifm_method=0)      # if_match_method absent from [IF-MATCH]: falls back to 0, not to the current 1 (H:611)
    for k, v in want.items():
        assert got[k] == v, (k, got[k], v)


def test_absent_keys_and_skipped_groups_keep_the_current_values(tmp_path):
    f = tmp_path / "vo.ini"
    f.write_text("[DETECT]\norb_nfeats = 777\n[LEAST_SQUARES]\nmax_iters = 5\n")
    ref = hip.default_params()
    p = hip.default_params()
    p.ifm_method = 1
    hip.load_params_ini(f, ["", "DETECT", "", "", "", "", ""], p)        # only the DETECT group is read
    assert p.orb_nfeats == 777 and p.max_iters == ref.max_iters and p.ifm_method == 1
    for k, _ in p._fields_:
        if k not in ("orb_nfeats", "ifm_method"):
            assert getattr(p, k) == getattr(ref, k), k
    # a named section that is not in the file leaves its group alone too, except if_match_method's hard 0
    hip.load_params_ini(f, ["R", "D", "M", "IFM", "LS", "G", "GEN"], p)
    assert p.orb_nfeats == 777 and p.ifm_method == 0


def test_missing_file_and_wrong_section_count(tmp_path):
    with pytest.raises(hip.SvoError):
        hip.load_params_ini(tmp_path / "nope.ini", SECTIONS)
    with pytest.raises(ValueError):
        hip.load_params_ini(tmp_path / "nope.ini", SECTIONS[:6])
    assert hip.lib().svo_params_load_ini(None, None, None) == -2
