"""Constants of R/common/h36m_dataset.py that the hot path reads: the 32->16 joint table (:37-38), the subject
lists and the Human3.6M camera calibration tables (:46-234, stored as numeric data in h36m_cameras.json)."""
import json
import os

H36M_32_To_16_Table = [0, 1, 2, 3, 6, 7, 8, 12, 13, 15, 17, 18, 19, 25, 26, 27]
TRAIN_SUBJECTS = ['S1', 'S5', 'S6', 'S7', 'S8']
TEST_SUBJECTS = ['S9', 'S11']

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "h36m_cameras.json")) as _f:
    _d = json.load(_f)
h36m_cameras_extrinsic_params = _d["extrinsic"]      # subject -> 4 x {orientation (w,x,y,z), translation (mm)}
h36m_cameras_intrinsic_params = _d["intrinsic"]      # 4 x {center, focal_length, radial/tangential distortion, res_w/h}
