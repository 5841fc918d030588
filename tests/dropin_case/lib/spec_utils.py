"""Stands for the checkout's lib/spec_utils.py: only names outside the hot path are ever taken from here (module __getattr__ of
the shadow)."""


def spectrogram_to_image(spec, mode='magnitude'):
    return 'checkout spectrogram_to_image'
