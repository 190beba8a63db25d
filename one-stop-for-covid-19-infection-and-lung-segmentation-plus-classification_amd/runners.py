"""The reference's task runners for the U-Net hot path, same names and observable outputs:

  holdout_runner_unet_infection_segmentation()   task1_preprocessing_plus_unet_with_comments.py:6
  runner_lung_segmentation()                      task3_lung_segmentation_unet.py:6

Zero positional arguments (app.py:45, 57 call them bare).  Data acquisition (pip / Kaggle /
Drive, T1:8-136) and NIfTI pre-processing (T1:163-686) are replaced by injectable arrays:
keyword ``data=(cts, masks)``, env ``UNET_DATA_NPZ=<file with x,y>``, else a synthetic set.
Literals of the reference are the defaults (new_dim 224 T1:479, batch 32 / 80 epochs T1:1041-1042,
lr 5e-4 T1:1053, test_size .3 / random_state 42 T1:762); env UNET_SIZE / UNET_EPOCHS /
UNET_SAMPLES / UNET_BATCH override them for smoke runs.  Each runner additionally RETURNS what
it printed, for tests.
"""
from __future__ import annotations

import os

import numpy as np

from .data import synthetic_ct, train_test_split
from .keras_like import UNetModel


def _env_int(name, default):
    v = os.environ.get(name)
    return int(v) if v else default


def _get_data(data, size, n_samples, seed):
    if data is not None:
        x, y = data
    elif os.environ.get("UNET_DATA_NPZ"):
        z = np.load(os.environ["UNET_DATA_NPZ"]); x, y = z["x"], z["y"]
    else:
        x, y = synthetic_ct(n_samples, size, seed)
    x = np.asarray(x, np.float32); y = np.asarray(y, np.float32)
    if x.ndim == 3:
        x, y = x[..., None], y[..., None]
    return x, y


def _segmentation_runner(tag, ckpt_dice, ckpt_loss, fine_range, data=None, input_size=None, epochs=None, batch_size=None,
                         n_samples=None, seed=0, backend=None, dropout=True, init_weights=None, workdir=".", verbose=1, **backend_kw):
    size = input_size or _env_int("UNET_SIZE", 224)
    epochs = epochs if epochs is not None else _env_int("UNET_EPOCHS", 80)
    batch_size = batch_size or _env_int("UNET_BATCH", 32)
    n_samples = n_samples or _env_int("UNET_SAMPLES", 64)
    cts, masks = _get_data(data, size, n_samples, seed)
    size = cts.shape[1]
    x_train, x_valid, y_train, y_valid = train_test_split(cts, masks, test_size=0.3, random_state=42)      # T1:762
    print(x_train.shape, x_valid.shape)                                                                     # T1:768
    model = UNetModel(size, cts.shape[-1], backend=backend, seed=seed, **backend_kw)                        # T1:853-915
    model.verbose = verbose
    if init_weights is not None:
        model.set_weights(init_weights)
    model.compile(lr=0.0005)                                                                                # T1:1053
    fd, fl = os.path.join(workdir, ckpt_dice), os.path.join(workdir, ckpt_loss)
    results = model.fit(x_train, y_train, batch_size=batch_size, epochs=epochs, validation_data=(x_valid, y_valid),
                        checkpoint_dice=fd, checkpoint_loss=fl, dropout=dropout, shuffle_seed=seed)         # T1:1059-1061
    if os.path.exists(fd):
        model.load_weights(fd)                                                                              # T1:1073
    out = {"history": results.history, "model": model, "tag": tag}
    score = model.evaluate(x_valid, y_valid, batch_size=32)                                                 # T1:1101
    score = [score["loss"], score["dice_coeff"]]
    print("test loss, test dice coefficient:", score)                                                       # T1:1102
    out["score"] = score

    the_range = np.arange(0.10, 0.80, 0.05)                                                                 # T1:1196
    ev = model.evaluate(x_valid, y_valid, batch_size=32, thresholds=the_range)                              # T1:1205-1211
    dices, ious = list(ev["dice"]), list(ev["iou"])
    print('DICES:', dices); print("IOUS:", ious)                                                            # T1:1217-1221
    print("Best Threshold:", the_range[np.argmax(dices)])
    print("Best dice score:", dices[np.argmax(dices)]); print("Best iou score:", ious[np.argmax(ious)])
    out.update(range=the_range, dices=dices, ious=ious)

    the_new_range = fine_range                                                                              # T1:1250 / T3:1206
    ev = model.evaluate(x_valid, y_valid, batch_size=32, thresholds=the_new_range)                          # T1:1259-1265
    new_dices, new_ious = list(ev["dice"]), list(ev["iou"])
    print("We just checked for", len(the_new_range), "steps between", round(float(the_new_range[0]), 2), "and",
          round(float(the_new_range[-1]) + 0.001, 2))                                                       # T1:1271
    print('NEW DICES:', new_dices); print("NEW IOUS:", new_ious)                                            # T1:1277-1281
    print("New Best Threshold:", the_new_range[np.argmax(new_dices)])
    print("Best new dice score:", new_dices[np.argmax(new_dices)]); print("Best new iou score:", new_ious[np.argmax(new_ious)])
    out.update(new_range=the_new_range, new_dices=new_dices, new_ious=new_ious)

    the_prec_rec_range = np.arange(0, 1, 0.05)                                                              # T1:1304
    ev = model.evaluate(x_valid, y_valid, batch_size=32, thresholds=the_prec_rec_range)                     # T1:1313-1319
    precisions, recalls = list(ev["precision"]), list(ev["recall"])
    print('PRECISIONS:', precisions); print("RRECALLS:", recalls)                                           # T1:1325-1330 (sic)
    print("Best Threshold for Precision:", the_prec_rec_range[np.argmax(precisions)])
    print("Best Threshold for Recall:", the_prec_rec_range[np.argmax(recalls)])
    print("Best precision score:", precisions[np.argmax(precisions)]); print("Best recall score:", recalls[np.argmax(recalls)])
    out.update(prec_rec_range=the_prec_rec_range, precisions=precisions, recalls=recalls)
    return out


def holdout_runner_unet_infection_segmentation(**kw):
    """Task 1 hold-out U-Net (app.py 'three').  Checkpoint names T1:1044-1045; fine sweep .52-.60 T1:1250."""
    return _segmentation_runner("infection", "unet_covid_weights_dice_coeff.hdf5", "unet_covid_weights_val_loss.hdf5",
                                np.arange(0.52, 0.60, 0.001), **kw)


def runner_lung_segmentation(**kw):
    """Task 3 lung U-Net (app.py 'six').  Same graph/recipe (T3:850-1009); fine sweep .43-.53 T3:1206;
    the checkpoint file names are the same "unet_covid_weights_*" literals, T3:991-992."""
    kw.setdefault("seed", 1)
    return _segmentation_runner("lung", "unet_covid_weights_dice_coeff.hdf5", "unet_covid_weights_val_loss.hdf5",
                                np.arange(0.43, 0.53, 0.001), **kw)
