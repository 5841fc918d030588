"""MI355X-native hot path of tsurumeso/vocal-remover behind the reference's Python API.

    nets.CascadedNet            <- lib/nets.py:44-141
    spec_utils.wave_to_spectrogram / spectrogram_to_wave / crop_center  <- lib/spec_utils.py
    dataset.make_padding        <- lib/dataset.py:198-205
    inference.Separator         <- inference.py:16-102
    train.train_epoch / validate_epoch  <- train.py:68-134

All math runs in libvr_mi355.so (hand-written HIP for gfx950) through a C ABI (include/vr_mi355.h);
there is no CPU / PyTorch fallback.  `dropin/` holds a `lib` package + `inference.py` that shadow
the reference's modules so its scripts run unchanged.
"""
from . import native  # noqa: F401
from . import audio, dataset, inference, nets, spec_utils  # noqa: F401

__all__ = ['native', 'nets', 'spec_utils', 'dataset', 'inference', 'audio']
