"""Audio-tagging / sound-event-detection evaluation (reference pytorch/evaluate.py:12-89).

`Evaluator.evaluate` runs the eval-mode HIP forward over a loader (pytorch_utils.forward), scores clipwise and
framewise average precision with scikit-learn, converts framewise probabilities to events (utils/vad.py through
utilities.frame_prediction_to_event_prediction), writes the submission file and scores it with the segment-based
metrics restated in utilities.segment_based_metrics (sed_eval, which the reference calls, is not installed).
"""
import numpy as np
from sklearn import metrics

from ..utils import config
from ..utils.utilities import frame_prediction_to_event_prediction, official_evaluate, write_submission
from .pytorch_utils import forward


def sed_average_precision(strong_target, framewise_output, average):
    """Framewise mAP over all (clip, frame) rows (evaluate.py:12-30).  average: None | 'macro' | 'micro'."""
    assert strong_target.shape == framewise_output.shape
    N, time_steps, classes_num = strong_target.shape
    return metrics.average_precision_score(strong_target.reshape((N * time_steps, classes_num)),
                                           framewise_output.reshape((N * time_steps, classes_num)), average=average)


class Evaluator(object):
    def __init__(self, model):
        self.model = model
        self.labels = config.labels
        self.idx_to_lb = config.idx_to_lb
        # default post-processing parameters of the reference (evaluate.py:45-50)
        self.sed_params_dict = {'audio_tagging_threshold': 0.5, 'sed_high_threshold': 0.5, 'sed_low_threshold': 0.2,
                                'n_smooth': 10, 'n_salt': 10}

    def evaluate(self, data_loader, reference_csv_path, submission_path):
        """-> (statistics, output_dict); statistics has 'clipwise_ap', 'framewise_ap' (when the pack carries strong
        labels) and 'sed_metrics' (evaluate.py:52-89)."""
        output_dict = forward(model=self.model, data_loader=data_loader, return_input=False, return_target=True)
        statistics = {'clipwise_ap': metrics.average_precision_score(output_dict['target'], output_dict['clipwise_output'],
                                                                     average=None)}
        if 'strong_target' in output_dict:
            # The packed strong labels have 1001 frames (utils/features.py:193) while every model emits 1000 (the 2x2
            # pooling floors 1001 -> 500, x8 interpolation): the reference's assert (evaluate.py:19) therefore fails on
            # real packs.  Decision here: score the frames both sides have, i.e. drop the trailing label frame(s).
            frames = min(output_dict['strong_target'].shape[1], output_dict['framewise_output'].shape[1])
            statistics['framewise_ap'] = sed_average_precision(output_dict['strong_target'][:, :frames],
                                                               output_dict['framewise_output'][:, :frames], average=None)
        predict_event_list = frame_prediction_to_event_prediction(output_dict, self.sed_params_dict)
        write_submission(predict_event_list, submission_path)
        statistics['sed_metrics'] = official_evaluate(reference_csv_path, submission_path)
        return statistics, output_dict
