"""A stand-in for the reference's menu script on machines where /root/reference does not exist (the GPU box): the same contract as
Scripts/app.py:7-57 -- star-import the six task modules, read one word from stdin, call the matching zero-argument runner -- written
independently (table dispatch) so the -m gpu drop-in test has something to feed to dropin/run_app.py.  Test infrastructure only."""
from task1_crossval_3folds_unet import *                     # noqa: F401,F403
from task1_crossval_4folds_unet import *                     # noqa: F401,F403
from task1_preprocessing_plus_unet_with_comments import *    # noqa: F401,F403
from task1_unet_plus_plus import *                           # noqa: F401,F403
from task2_covid19_classifcation import *                    # noqa: F401,F403
from task3_lung_segmentation_unet import *                   # noqa: F401,F403

CHOICES = {"one": "three_fold_runner_unet_infection_segmentation", "two": "four_fold_runner_unet_infection_segmentation",
           "three": "holdout_runner_unet_infection_segmentation", "four": "holdout_runner_unetplusplus_infection_segmentation",
           "five": "runner_classification", "six": "runner_lung_segmentation"}
word = input().strip()
if word in CHOICES:
    globals()[CHOICES[word]]()
