"""Drop-in for the reference's utils/utils.py: same function, same arguments, MI355X extractor.

Reference: utils/utils.py:6-26 -- `get_feat_extractor(num_samples, num_filters, use_kaldi=False)` always
returns `Fbank(FbankConfig(num_filters=num_filters, frame_shift=1/num_samples))` (line 25 overrides the
kaldifeat branch).  Here the returned object computes the same features on the GPU (feats.HipFbank).
"""
import os
import sys

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

from feats import HipFbank, HipFbankConfig  # noqa: E402


def get_feat_extractor(num_samples, num_filters, use_kaldi=False):
    """Feature extractor producing `num_samples` frames per second with `num_filters` mel bins.

    `use_kaldi` is accepted for signature compatibility; as in the reference it does not change the result.
    """
    frame_shift = 1 / num_samples
    return HipFbank(HipFbankConfig(num_filters=num_filters, frame_shift=frame_shift))
