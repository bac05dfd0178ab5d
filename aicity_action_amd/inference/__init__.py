from .sliding_window import SlidingWindowClassifier, frame_idxs_uniform, get_proposals  # noqa: F401
from .postprocess import (aggregate_predictions, compute_f1, get_chunks, merge_views, video_action_chunks,  # noqa: F401
                          write_submission)
