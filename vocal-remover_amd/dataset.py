"""Host-side crop geometry (the only piece of lib/dataset.py on the hot path)."""


def make_padding(width, cropsize, offset):
    """dataset.make_padding (lib/dataset.py:198-205): left pad, right pad, roi size.

    Note the reference adds a full extra roi on the right when width % roi == 0; kept.
    """
    left = offset
    roi_size = cropsize - offset * 2
    if roi_size == 0:
        roi_size = cropsize
    right = roi_size - (width % roi_size) + left
    return left, right, roi_size
