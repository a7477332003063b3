"""Host-side input feeds with the reference's contract (ultra.input_layer.*): they build the `input_feed` dict
(`letor_features` [n_docs,F], `docid_input{l}` [B] float32, `label{l}` [B] float32; pad id = n_docs)."""
from .base_input_feed import BaseInputFeed  # noqa: F401
from .click_simulation_feed import ClickSimulationFeed  # noqa: F401
from .direct_label_feed import DirectLabelFeed  # noqa: F401
from .device_click_feed import DeviceClickFeed, ResidentDataset  # noqa: F401
