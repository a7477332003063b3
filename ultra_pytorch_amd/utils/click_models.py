"""Click simulators (host side): position-biased and cascade models driven by Python's `random` stream, so that a
seeded run draws the same clicks as the reference's simulators (ultra/utils/click_models.py:7-16, 68-110, 187-236):
one `random.random()` per list position, click iff u < exam_prob(rank) * click_prob[label]."""
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


def loadModelFromJson(model_desc):
    name = model_desc["model_name"]
    if name == "user_browsing_model":
        raise NotImplementedError("user_browsing_model is not provided by this engine (PBM and cascade are)")
    model = CascadeModel() if name == "cascade_model" else PositionBiasedModel()
    model.eta = model_desc["eta"]
    model.click_prob = model_desc["click_prob"]
    model.exam_prob = model_desc["exam_prob"]
    return model
