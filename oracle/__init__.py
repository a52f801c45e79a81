"""Test infrastructure only: CPU restatement (oracle) of the reference's EdgeConv hot path.
Never imported by the product package dynamic-gcnn_amd/dgcnn."""
