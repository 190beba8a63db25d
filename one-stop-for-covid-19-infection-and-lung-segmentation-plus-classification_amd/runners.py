"""The reference's task runners for the U-Net hot path, same names and observable outputs:

  holdout_runner_unet_infection_segmentation()   task1_preprocessing_plus_unet_with_comments.py:6
  runner_lung_segmentation()                      task3_lung_segmentation_unet.py:6
  three_/four_fold_runner_unet_infection_segmentation(), holdout_runner_unetplusplus_infection_segmentation(), runner_classification()

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

from . import dp_launch
from .data import kfold_indices, synthetic_classification, synthetic_ct, train_test_split
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
                         n_samples=None, seed=0, backend=None, dropout=True, init_weights=None, workdir=".", verbose=1, arch="unet",
                         **backend_kw):
    size = input_size or _env_int("UNET_SIZE", 224)
    epochs = epochs if epochs is not None else _env_int("UNET_EPOCHS", 80)
    batch_size = batch_size or _env_int("UNET_BATCH", 32)
    n_samples = n_samples or _env_int("UNET_SAMPLES", 64)
    cts, masks = _get_data(data, size, n_samples, seed)
    size = cts.shape[1]
    x_train, x_valid, y_train, y_valid = train_test_split(cts, masks, test_size=0.3, random_state=42)      # T1:762
    print(x_train.shape, x_valid.shape)                                                                     # T1:768
    model = UNetModel(size, cts.shape[-1], backend=backend, seed=seed, arch=arch, **backend_kw)             # T1:853-915
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
    """Task 1 hold-out U-Net (app.py 'three').  Checkpoint names T1:1044-1045; fine sweep .52-.60 T1:1250.
    UNET_GPUS=N in the environment: data parallel on N GPUs (dp_launch.py)."""
    r = dp_launch.maybe_launch("holdout_runner_unet_infection_segmentation", kw)
    if r is not None:
        return r
    return _segmentation_runner("infection", "unet_covid_weights_dice_coeff.hdf5", "unet_covid_weights_val_loss.hdf5",
                                np.arange(0.52, 0.60, 0.001), **kw)


def runner_lung_segmentation(**kw):
    """Task 3 lung U-Net (app.py 'six').  Same graph/recipe (T3:850-1009); fine sweep .43-.53 T3:1206;
    the checkpoint file names are the same "unet_covid_weights_*" literals, T3:991-992.  UNET_GPUS=N in the environment: data parallel on
    N GPUs, the global batch (32, or UNET_BATCH=64 for BASELINE.json configs[2]) sharded over the ranks (dp_launch.py)."""
    kw.setdefault("seed", 1)
    r = dp_launch.maybe_launch("runner_lung_segmentation", kw)
    if r is not None:
        return r
    return _segmentation_runner("lung", "unet_covid_weights_dice_coeff.hdf5", "unet_covid_weights_val_loss.hdf5",
                                np.arange(0.43, 0.53, 0.001), **kw)


def _kfold_runner(k, data=None, input_size=None, epochs=None, batch_size=None, n_samples=None, seed=0, backend=None, dropout=True,
                  init_weights=None, workdir=".", verbose=1, reinit_each_fold=False, overwrite_fold_files=True, **backend_kw):
    """K-fold driver of task1_crossval_{3,4}folds_unet.py (CV4:1045-1108, 1183-1330; CV3:1005-1052, 1136-1300): same U-Net,
    same recipe, KFold(n_splits=k, random_state=42, shuffle=True).  Faithful to the reference by default, including its two
    quirks: ONE model object is trained through all folds without re-initialisation (CV4:1019, 1051-1090 -> fold leakage,
    `reinit_each_fold=False`), and after training the FINAL weights overwrite every fold's best-checkpoint file
    (CV4:1105-1108, `overwrite_fold_files=True`).  Set the flags the other way for a sound cross-validation."""
    import time
    size = input_size or _env_int("UNET_SIZE", 224)
    epochs = epochs if epochs is not None else _env_int("UNET_EPOCHS", 80)
    batch_size = batch_size or _env_int("UNET_BATCH", 32)
    n_samples = n_samples or _env_int("UNET_SAMPLES", 64)
    cts, infections = _get_data(data, size, n_samples, seed)
    size = cts.shape[1]
    model = UNetModel(size, cts.shape[-1], backend=backend, seed=seed, **backend_kw)
    model.verbose = verbose
    w0 = init_weights if init_weights is not None else model.get_weights()
    model.set_weights(w0)
    paths = [os.path.join(workdir, f"unet_covid_fold{i + 1}.hdf5") for i in range(k)]                       # CV4:1029-1032
    folds = kfold_indices(len(cts), k, 42)                                                                   # CV4:1047
    bar = "%" * 180
    start = time.perf_counter()
    histories = []
    for fold_number, (train_index, test_index) in enumerate(folds, 1):
        print(bar); print("Current fold number going:", fold_number); print(bar)                             # CV4:1053-1055
        x_train, x_valid = cts[train_index], cts[test_index]
        y_train, y_valid = infections[train_index], infections[test_index]
        print("Shapes:", x_train.shape, x_valid.shape)                                                       # CV4:1059
        if reinit_each_fold:
            model.set_weights(w0)
        model.compile(lr=0.0005)                                                                             # CV4:1062
        histories.append(model.fit(x_train, y_train, batch_size=batch_size, epochs=epochs, validation_data=(x_valid, y_valid),
                                   checkpoint_dice=paths[fold_number - 1], dropout=dropout, shuffle_seed=seed + fold_number).history)
    print(f"Time of {k}-fold cross validation: ", time.perf_counter() - start)                               # CV4:1099
    from .keras_like import dp_info
    world, rank = dp_info(model.backend)
    if overwrite_fold_files:
        for p in paths:
            if rank == 0:
                model.save_weights(p)                                                                        # CV4:1105-1108
        if world > 1:
            model.backend.barrier()
    out = {"histories": histories, "paths": paths, "folds": folds, "model": model, "scores": []}
    dots = "." * 118
    for split_number, (train_index, test_index) in enumerate(folds, 1):                                      # CV4:1183-1195
        print(dots); print("Current fold number going:", split_number); print(dots)
        model.load_weights(paths[split_number - 1])
        ev = model.evaluate(cts[test_index], infections[test_index], batch_size=32)
        score = [ev["loss"], ev["dice_coeff"]]
        print("test loss, test dice coefficient:", score)
        out["scores"].append(score)
    the_range = np.arange(0.30, 0.80, 0.05)                                                                  # CV4:1222
    print(len(the_range))
    tables = {m: [] for m in ("dice", "iou", "precision", "recall")}
    for split_number, (train_index, test_index) in enumerate(folds, 1):                                      # CV4:1231-1266
        print("." * 145); print("Current split number going:", split_number); print("." * 145)
        model.load_weights(paths[split_number - 1])
        for t in the_range:
            print("%" * 73); print("Calculating for threshold:", t); print("%" * 73)
        ev = model.evaluate(cts[test_index], infections[test_index], batch_size=32, thresholds=the_range)    # one pass for all t
        for m in tables:
            tables[m].append(np.asarray(ev[m]))
    names = {"dice": ("Dices", "dice", "dices"), "iou": ("Ious", "iou", "ious"), "precision": ("precision", "precision", "precisions"),
             "recall": ("recall", "recall", "recalls")}
    for m, (title, one, many) in names.items():                                                              # CV4:1270-1365
        tab = np.transpose(np.array(tables[m]))                       # rows: thresholds, columns: split number
        out["table_" + m] = tab
        print(f"{k}-fold {title} dataframe"); print("Rows indices: Thresholds, Column indices: Split Number")
        print(f"Maximum validation {one} on any splits:", np.max(tab))
        print(f"Maximum validation {one} on each of the {k} splits (any threshold chosen):", tab.max(axis=0))
        print("Best threshold for each split", *[the_range[int(np.argmax(tab[:, j]))] for j in range(k)])
        print(f"Mean of all obtained {many}:", tab.mean(axis=0).mean())
    out["range"] = the_range
    return out


def three_fold_runner_unet_infection_segmentation(**kw):
    """Task 1, 3-fold cross-validation U-Net (app.py 'one'; task1_crossval_3folds_unet.py:6)."""
    r = dp_launch.maybe_launch("three_fold_runner_unet_infection_segmentation", kw)
    if r is not None:
        return r
    return _kfold_runner(3, **kw)


def four_fold_runner_unet_infection_segmentation(**kw):
    """Task 1, 4-fold cross-validation U-Net (app.py 'two'; task1_crossval_4folds_unet.py:6)."""
    r = dp_launch.maybe_launch("four_fold_runner_unet_infection_segmentation", kw)
    if r is not None:
        return r
    return _kfold_runner(4, **kw)


def holdout_runner_unetplusplus_infection_segmentation(**kw):
    """Task 1 hold-out U-Net++ (app.py 'four'; task1_unet_plus_plus.py:6): nested-skip graph UPP:858-950, same recipe
    (Adam 5e-4, bce_dice_loss, batch 32, 80 epochs UPP:1054-1074), fine sweep .40-.50 (UPP:1274)."""
    kw.setdefault("seed", 2)
    r = dp_launch.maybe_launch("holdout_runner_unetplusplus_infection_segmentation", kw)
    if r is not None:
        return r
    return _segmentation_runner("infection_unetpp", "unet_covid_weights_dice_coeff.hdf5", "unet_covid_weights_val_loss.hdf5",
                                np.arange(0.40, 0.50, 0.001), arch="unetpp", **kw)


def runner_classification(data=None, input_size=None, epochs=None, batch_size=None, n_samples=None, seed=3, backend=None, dropout=True,
                          init_weights=None, workdir=".", verbose=1, honour_array_class_weight=False, **backend_kw):
    """Task 2 slice classifier (app.py 'five'; task2_covid19_classifcation.py:6): infected / not-infected CT slice.
    StratifiedShuffleSplit(1, .3, 42) T2:647; CNN T2:747-776; balanced class weights T2:801 (printed; passed to fit as the ndarray the
    reference passes, which Keras 2.3 ignores -- `honour_array_class_weight=True` applies them); batch 32 / 25 epochs T2:811-812;
    RocCallback + ModelCheckpoint(val_loss) T2:814-820; best-AUC weights reloaded T2:851; evaluate T2:884; confusion-matrix
    reports at thresholds 0.50 and 0.81 T2:916-967.  `data=(cts [N,S,S,1], y_label [N])` or env UNET_DATA_NPZ, else synthetic."""
    from .classifier import ClassifierModel, compute_class_weight_balanced, confusion_report, stratified_shuffle_split
    _kw = dict(data=data, input_size=input_size, epochs=epochs, batch_size=batch_size, n_samples=n_samples, seed=seed, backend=backend, dropout=dropout,
               init_weights=init_weights, workdir=workdir, verbose=verbose, honour_array_class_weight=honour_array_class_weight, **backend_kw)
    r = dp_launch.maybe_launch("runner_classification", {k: v for k, v in _kw.items() if v is not None})      # UNET_GPUS=N: data parallel (dp_launch.py)
    if r is not None:
        return r
    size = input_size or _env_int("UNET_SIZE", 224)
    epochs = epochs if epochs is not None else _env_int("UNET_EPOCHS", 25)
    batch_size = batch_size or _env_int("UNET_BATCH", 32)
    n_samples = n_samples or _env_int("UNET_SAMPLES", 64)
    if data is not None:
        cts, y_label = data
    elif os.environ.get("UNET_DATA_NPZ"):
        z = np.load(os.environ["UNET_DATA_NPZ"]); cts, y_label = z["x"], z["y"]
    else:
        cts, y_label = synthetic_classification(n_samples, size, seed)
    cts = np.asarray(cts, np.float32); y_label = np.asarray(y_label).reshape(-1).astype(np.int64)
    if cts.ndim == 3:
        cts = cts[..., None]
    size = cts.shape[1]
    print(cts.shape, y_label.shape)                                                                          # T2:512
    train_index, test_index = stratified_shuffle_split(y_label, 0.3, 42)                                     # T2:647-650
    x_train, x_valid = cts[train_index], cts[test_index]
    y_train, y_valid = y_label[train_index], y_label[test_index]
    print(x_train.shape, x_valid.shape); print(y_train.shape, y_valid.shape)                                 # T2:675-676
    model = ClassifierModel(size, cts.shape[-1], backend=backend, seed=seed, **backend_kw)                   # T2:747-776
    model.verbose = verbose
    if init_weights is not None:
        model.set_weights(init_weights)
    if verbose:
        model.summary()                                                                                      # T2:778
    weights = compute_class_weight_balanced(y_train)                                                         # T2:801-803
    print(weights)                                                                                           # T2:804
    fl, fa = os.path.join(workdir, "covid_weights_val_loss.hdf5"), os.path.join(workdir, "best_val_auc_weights.h5")   # T2:818, 733
    model.compile(lr=0.0005)                                                                                 # T2:829
    results = model.fit(x_train, y_train, batch_size=batch_size, epochs=epochs, validation_data=(x_valid, y_valid), class_weight=weights,
                        honour_array_class_weight=honour_array_class_weight, best_auc_path=fa, checkpoint_loss=fl, dropout=dropout,
                        shuffle_seed=seed)                                                                   # T2:833-835
    if os.path.exists(fa):
        model.load_weights(fa)                                                                               # T2:851
    from .keras_like import dp_info
    if dp_info(model.backend)[1] == 0:
        with open(os.path.join(workdir, "best_val_auc_weights.json"), "w") as f:                             # T2:867-869
            f.write(model.to_json())
    print("Best saved AUCROC on validation set :", model.best_val_auc)                                       # T2:878
    score = model.evaluate(x_valid, y_valid, batch_size=32)                                                  # T2:884
    print("test loss:", score[0], "\ntest f1 score:", score[1])                                              # T2:885
    predictions = np.array(model.predict(x_valid).flatten())                                                 # T2:910-911
    out = {"history": results.history, "model": model, "score": score, "predictions": predictions, "y_valid": y_valid,
           "class_weights": weights, "best_val_auc": model.best_val_auc, "reports": {}}
    for thr in (0.50, 0.81):                                                                                 # T2:916, 946
        r = confusion_report(y_valid, predictions, thr)
        print('Accuracy:', r["accuracy"]); print('Precision:', r["precision"]); print('Recall:', r["recall"])   # T2:936-939
        print('F1 score:', r["f1"])
        out["reports"][thr] = r
    return out
