"""Click simulators (host side): position-biased, user-browsing and cascade models driven by Python's `random` stream, so that
a seeded run draws the same clicks as the reference's simulators (ultra/utils/click_models.py:7-16, 68-110, 113-186, 187-236):
one `random.random()` per list position, click iff u < exam_prob * click_prob[label]."""
import random


class ClickModel(object):
    model_name = "click_model"

    def __init__(self, neg_click_prob=0.0, pos_click_prob=1.0, relevance_grading_num=1, eta=1.0):
        self.exam_prob = None
        self.setExamProb(eta)
        b = (pos_click_prob - neg_click_prob) / (pow(2, relevance_grading_num) - 1)
        a = neg_click_prob - b
        self.click_prob = [a + pow(2, i) * b for i in range(relevance_grading_num + 1)]

    def setExamProb(self, eta):
        self.eta = eta

    def getModelJson(self):
        return {"model_name": self.model_name, "eta": self.eta, "click_prob": self.click_prob, "exam_prob": self.exam_prob}

    def getExamProb(self, rank):
        return self.exam_prob[rank if rank < len(self.exam_prob) else -1]

    def sampleClick(self, rank, relevance_label):
        relevance_label = int(relevance_label) if relevance_label > 0 else 0
        exam_p = self.getExamProb(rank)
        click_p = self.click_prob[relevance_label if relevance_label < len(self.click_prob) else -1]
        return (1 if random.random() < exam_p * click_p else 0), exam_p, click_p


class PositionBiasedModel(ClickModel):
    model_name = "position_biased_model"
    ORIGINAL_EXAM_PROB = [0.68, 0.61, 0.48, 0.34, 0.28, 0.20, 0.11, 0.10, 0.08, 0.06]

    def setExamProb(self, eta):
        self.eta = eta
        self.exam_prob = [pow(x, eta) for x in self.ORIGINAL_EXAM_PROB]

    def sampleClicksForOneList(self, label_list):
        out = [self.sampleClick(rank, label) for rank, label in enumerate(label_list)]
        return [o[0] for o in out], [o[1] for o in out], [o[2] for o in out]

    def estimatePropensityWeightsForOneList(self, click_list, use_non_clicked_data=False):
        return [(1.0 / self.getExamProb(r) * self.getExamProb(0)) if (use_non_clicked_data or c > 0) else 0.0
                for r, c in enumerate(click_list)]


class UserBrowsingModel(ClickModel):
    """Examination depends on the rank AND on how far back the last click was (reference click_models.py:113-186): a triangular
    table exam[rank][distance - 1], distance = rank - last_click_rank (last_click_rank = -1 before the first click).  Beyond the
    table's last row the last row is reused: a distance that reaches back before the list start takes its last entry ("no click
    so far"), any other distance its own column, saturating at the second-to-last."""
    model_name = "user_browsing_model"
    RD_EXAM_TABLE = [
        [1.0],
        [0.98, 1.0],
        [1.0, 0.62, 0.95],
        [1.0, 0.77, 0.42, 0.82],
        [1.0, 0.92, 0.55, 0.31, 0.69],
        [1.0, 0.96, 0.63, 0.4, 0.22, 0.54],
        [1.0, 0.99, 0.73, 0.46, 0.29, 0.17, 0.47],
        [1.0, 1.0, 0.89, 0.52, 0.35, 0.24, 0.14, 0.43],
        [1.0, 1.0, 0.95, 0.68, 0.4, 0.29, 0.19, 0.12, 0.41],
        [1.0, 1.0, 1.0, 0.96, 0.52, 0.36, 0.27, 0.18, 0.12, 0.43],
    ]

    def setExamProb(self, eta):
        self.eta = eta
        self.exam_prob = [[pow(x, eta) for x in row] for row in self.RD_EXAM_TABLE]

    def getExamProb(self, rank, last_click_rank=-1):
        distance = rank - last_click_rank
        if rank < len(self.exam_prob):
            return self.exam_prob[rank][distance - 1]
        last = self.exam_prob[-1]
        if distance > rank:
            return last[-1]
        return last[distance - 1 if distance < len(last) - 1 else -2]

    def sampleClick(self, rank, last_click_rank, relevance_label):
        relevance_label = int(relevance_label) if relevance_label > 0 else 0
        exam_p = self.getExamProb(rank, last_click_rank)
        click_p = self.click_prob[relevance_label if relevance_label < len(self.click_prob) else -1]
        return (1 if random.random() < exam_p * click_p else 0), exam_p, click_p

    def sampleClicksForOneList(self, label_list):
        clicks, exams, cps, last = [], [], [], -1
        for rank, label in enumerate(label_list):
            click, exam_p, click_p = self.sampleClick(rank, last, label)
            last = rank if click > 0 else last
            clicks.append(click)
            exams.append(exam_p)
            cps.append(click_p)
        return clicks, exams, cps

    def estimatePropensityWeightsForOneList(self, click_list, use_non_clicked_data=False):
        out, last = [], -1
        for r, c in enumerate(click_list):
            out.append(1.0 / self.getExamProb(r, last) if (use_non_clicked_data or c > 0) else 0.0)
            last = r if c > 0 else last
        return out


class CascadeModel(ClickModel):
    model_name = "cascade_model"

    def setExamProb(self, eta):
        self.eta = eta
        self.exam_prob = [1.0 for _ in range(10)]

    def sampleClicksForOneList(self, label_list):
        clicks, exams, cps, has_click = [], [], [], False
        for rank, label in enumerate(label_list):
            click, exam_p, click_p = self.sampleClick(rank, label)  # the uniform is drawn even after a click
            clicks.append(0.0 if has_click else click)
            exams.append(0.0 if has_click else exam_p)
            cps.append(click_p)
            has_click = has_click or click > 0
        return clicks, exams, cps

    def estimatePropensityWeightsForOneList(self, click_list, use_non_clicked_data=False):
        return [(1.0 / self.getExamProb(r) * self.getExamProb(0)) if (use_non_clicked_data or c > 0) else 0.0
                for r, c in enumerate(click_list)]


def loadModelFromJson(model_desc):
    name = model_desc["model_name"]
    model = {"cascade_model": CascadeModel, "user_browsing_model": UserBrowsingModel}.get(name, PositionBiasedModel)()
    model.eta = model_desc["eta"]
    model.click_prob = model_desc["click_prob"]
    model.exam_prob = model_desc["exam_prob"]
    return model
