"""Training driver for the accelerated 1-d hierarchical CNN: counterpart of the reference's
train_hierarchical_cnn.py, which is its train_2d_cnn.py with `HierarchicalCNNClassificationModel`
(train_hierarchical_cnn.py:16,355) and the default label "1d_cnn" (:185) -- every flag and the whole per-fold flow
are shared (see train_2d_cnn.py in this directory)."""
from freesound_classification_amd.networks.classifiers import HierarchicalCNNClassificationModel
from train_2d_cnn import main

if __name__ == "__main__":
    main(HierarchicalCNNClassificationModel, default_label="1d_cnn")
