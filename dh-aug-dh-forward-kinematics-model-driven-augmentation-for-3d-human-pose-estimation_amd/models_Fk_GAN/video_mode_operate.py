"""Only the piece of R/models_Fk_GAN/video_mode_operate.py that the hot path reads."""


def video_receptive_field(filter_widths):
    """frames per sample = product of the temporal filter widths (R/models_Fk_GAN/video_mode_operate.py:411-415)."""
    frames = 1
    for w in filter_widths:
        frames *= w
    return frames


def frames_from_args(args):
    if getattr(args, "single_or_multi_train_mode", "single") == "multi":
        return video_receptive_field([int(x) for x in args.architecture.split(",")])
    return 1
