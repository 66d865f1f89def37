"""Confusion-matrix metrics of the reference's misc/metric_tool.py (same names, same formulas -- these are the published
definitions of accuracy / precision / recall / F1 / IoU with the reference's float32-eps regularisation):
AverageMeter, ConfuseMatrixMeter.update_cm / get_scores, cm2F1, cm2score, get_confuse_matrix, get_mIoU.

The hot evaluation loop does not come through here per batch: CDTrainer / CDEvaluator count on the device
(dh_confusion_matrix) and hand the accumulated matrix to cm2score; ConfuseMatrixMeter.update_cm(pr, gt) remains for
callers that hold numpy masks (and is what the fixture of tests/test_host_plumbing_cpu.py pins against the reference)."""
import numpy as np

_EPS = np.finfo(np.float32).eps


class AverageMeter:
    def __init__(self):
        self.initialized = False
        self.val = self.avg = self.sum = self.count = None

    def initialize(self, val, weight):
        self.val, self.avg, self.sum, self.count, self.initialized = val, val, val * weight, weight, True

    def update(self, val, weight=1):
        if not self.initialized:
            self.initialize(val, weight)
        else:
            self.add(val, weight)

    def add(self, val, weight):
        self.val = val
        self.sum += val * weight
        self.count += weight
        self.avg = self.sum / self.count

    def value(self):
        return self.val

    def average(self):
        return self.avg

    def get_scores(self):
        return cm2score(self.sum)

    def clear(self):
        self.initialized = False


class ConfuseMatrixMeter(AverageMeter):
    def __init__(self, n_class):
        super().__init__()
        self.n_class = n_class

    def update_cm(self, pr, gt, weight=1):
        """confusion matrix of this batch -> running sum; returns the batch's mean F1"""
        val = get_confuse_matrix(num_classes=self.n_class, label_gts=gt, label_preds=pr)
        self.update(val, weight)
        return cm2F1(val)

    def update_from_matrix(self, cm, weight=1):
        """the same with a confusion matrix counted elsewhere (the device kernel)"""
        val = np.asarray(cm, dtype=np.float64)
        self.update(val, weight)
        return cm2F1(val)

    def get_scores(self):
        return cm2score(self.sum)


def _prf(hist):
    tp = np.diag(hist)
    rows, cols = hist.sum(axis=1), hist.sum(axis=0)
    recall, precision = tp / (rows + _EPS), tp / (cols + _EPS)
    f1 = 2 * recall * precision / (recall + precision + _EPS)
    return tp, rows, cols, recall, precision, f1


def cm2F1(confusion_matrix):
    return np.nanmean(_prf(confusion_matrix)[5])


def cm2score(confusion_matrix):
    hist = confusion_matrix
    n_class = hist.shape[0]
    tp, rows, cols, recall, precision, f1 = _prf(hist)
    acc = tp.sum() / (hist.sum() + _EPS)
    iu = tp / (rows + cols - tp + _EPS)
    score = {'acc': acc, 'miou': np.nanmean(iu), 'mf1': np.nanmean(f1)}
    for name, vals in (('iou_', iu), ('F1_', f1), ('precision_', precision), ('recall_', recall)):
        score.update({name + str(i): vals[i] for i in range(n_class)})
    return score


def get_confuse_matrix(num_classes, label_gts, label_preds):
    cm = np.zeros((num_classes, num_classes))
    for lt, lp in zip(label_gts, label_preds):
        lt, lp = np.asarray(lt).flatten(), np.asarray(lp).flatten()
        keep = (lt >= 0) & (lt < num_classes)
        cm += np.bincount(num_classes * lt[keep].astype(int) + lp[keep], minlength=num_classes ** 2).reshape(num_classes, num_classes)
    return cm


def get_mIoU(num_classes, label_gts, label_preds):
    return cm2score(get_confuse_matrix(num_classes, label_gts, label_preds))['miou']
