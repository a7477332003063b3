"""Inverse-propensity tables (reference ultra/utils/propensity_estimator.py:10-56): a JSON file with "IPW_list"."""
import json


class BasicPropensityEstimator(object):
    def __init__(self, file_name=None):
        self.IPW_list = []
        if file_name:
            self.loadEstimatorFromFile(file_name)

    def getPropensityForOneList(self, click_list, use_non_clicked_data=False):
        """weight_r = IPW_list[min(r, len-1)] if clicked (or use_non_clicked_data) else 0   (:22-42)"""
        last = len(self.IPW_list) - 1
        return [self.IPW_list[min(r, last)] if (use_non_clicked_data or c > 0) else 0.0 for r, c in enumerate(click_list)]

    def loadEstimatorFromFile(self, file_name):
        with open(file_name) as f:
            self.IPW_list = json.load(f)["IPW_list"]

    def outputEstimatorToFile(self, file_name):
        with open(file_name, "w") as f:
            f.write(json.dumps({"IPW_list": self.IPW_list}, indent=4, sort_keys=True))


class RandomizedPropensityEstimator(BasicPropensityEstimator):
    """Loads a pre-estimated table; re-estimating one from 10^7 simulated sessions (:95-132) is offline tooling."""
