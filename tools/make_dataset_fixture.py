#!/usr/bin/env python3
"""Write tests/golden/dataset_fixture/may_face_crop_lip: a tiny clip in the reference's on-disk layout
(src/data/someones_lip_dataset.py:43-120) whose every value is chosen so that what the reference reader would yield can be
TRACED BY HAND (tests/test_data_reader.py::test_committed_fixture_hand_traced holds the traced numbers as literals):

  audio/audio.npy        float64 [20,16,29], audio[k] = k everywhere
  audio_test/audio.npy   float64 [5,16,29],  = 100 + k
  coords/%05d.npy        float32 [12,16,2],  = (k + 1) / 100
  ori_images_face/%05d.jpg  12x16 flat grey level 10 * (k + 1)      (flat images survive JPEG exactly)
  images/%05d.jpg           6x8 flat grey 200                        (defines the lip crop size 6 x 8)
  canonical_lip_mask.jpg    12x16, white box rows 4..9, cols 5..12
  landmarks/00001.lms       68 points; mouth points (48..67) span x in [5.5, 12.25], y in [4.75, 8.5]
Training-side inputs (someones_lip_dataset.py:75-93, 113-120, 328-392):
  audio/mel.npy          float32 [80,64], mel[c,t] = t + c/100: what `melspectrogram(load_wav(audio.wav))` would return -- the mel
                         front-end (src/data/audio.py, librosa) is out of scope, so the reader takes the spectrogram precomputed
  face_bbox_dict.npy     pickled dict {"%05d.jpg": float32 [x, y, x2, y2, conf]} (preprocess/detect_landmarks.py:33-63)
  track_params.pt        {'euler': [20,3], 'trans': [20,3]} float32, euler[k] = (k, k+0.25, k+0.5)/100, trans[k] = (k, -k, 10+k)/10
  canonical_head_mask.jpg / canonical_face_mask.jpg   12x16 white boxes rows 1..10 x cols 2..13 / rows 3..8 x cols 4..11

The decoded PIXELS are whatever PIL's libjpeg yields; the reference decodes with cv2.imread / imageio (not installed in this
image), so pixel-level agreement with the reference reader is NOT pinned -- only the conventions are (split slices, index and file
naming, the lip box from the landmarks, BGR channel order of the mask, float64 -> float32 audio)."""
import os

import numpy as np
from PIL import Image

ROOT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "dataset_fixture", "may_face_crop_lip")
N, FH, FW, LH, LW = 20, 12, 16, 6, 8
for sub in ("audio", "audio_test", "coords", "ori_images_face", "images", "landmarks"):
    os.makedirs(os.path.join(ROOT, sub), exist_ok=True)
np.save(os.path.join(ROOT, "audio", "audio.npy"), np.arange(N, dtype=np.float64)[:, None, None] * np.ones((1, 16, 29)))
np.save(os.path.join(ROOT, "audio_test", "audio.npy"), (100 + np.arange(5, dtype=np.float64))[:, None, None] * np.ones((1, 16, 29)))
for k in range(N):
    np.save(os.path.join(ROOT, "coords", "%05d.npy" % (k + 1)), np.full((FH, FW, 2), (k + 1) / 100.0, np.float32))
    Image.fromarray(np.full((FH, FW, 3), 10 * (k + 1), np.uint8)).save(os.path.join(ROOT, "ori_images_face", "%05d.jpg" % (k + 1)), quality=100)
    Image.fromarray(np.full((LH, LW, 3), 200, np.uint8)).save(os.path.join(ROOT, "images", "%05d.jpg" % (k + 1)), quality=100)
mask = np.zeros((FH, FW, 3), np.uint8)
mask[4:10, 5:13] = 255
Image.fromarray(mask).save(os.path.join(ROOT, "canonical_lip_mask.jpg"), quality=100)
lms = np.full((68, 2), 1.0, np.float32)
lms[48:, 0] = np.linspace(5.5, 12.25, 20)
lms[48:, 1] = np.linspace(4.75, 8.5, 20)
np.savetxt(os.path.join(ROOT, "landmarks", "00001.lms"), lms)
import torch  # noqa: E402

np.save(os.path.join(ROOT, "audio", "mel.npy"), (np.arange(64, dtype=np.float32)[None, :] + np.arange(80, dtype=np.float32)[:, None] / 100))
np.save(os.path.join(ROOT, "face_bbox_dict.npy"),
        {"%05d.jpg" % (k + 1): np.array([2 + k % 2, 1, 14, 11 - k % 3, 0.9 + k / 1000], np.float32) for k in range(N)})
kk = torch.arange(N, dtype=torch.float32)[:, None]
torch.save({"euler": (kk + torch.tensor([[0.0, 0.25, 0.5]])) / 100, "trans": torch.cat([kk, -kk, 10 + kk], 1) / 10},
           os.path.join(ROOT, "track_params.pt"))
for name, (r0, r1, c0, c1) in (("canonical_head_mask.jpg", (1, 11, 2, 14)), ("canonical_face_mask.jpg", (3, 9, 4, 12))):
    mk = np.zeros((FH, FW, 3), np.uint8)
    mk[r0:r1, c0:c1] = 255
    Image.fromarray(mk).save(os.path.join(ROOT, name), quality=100)
print("wrote", ROOT)
