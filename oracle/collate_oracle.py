"""CPU restatement of the reference's collate_fn (src/data_loader.py:59-122) for use_bert=False, line by line with torch's own
pad_sequence (TEST INFRASTRUCTURE: only tests/ may import anything under oracle/).  The reference module itself cannot be
imported offline (it fetches `bert-base-uncased` at import, data_loader.py:15), so this restatement is pinned by reading the
source only: parity unpinned by a run of the reference."""
import numpy as np
import torch
from torch.nn.utils.rnn import pad_sequence

PAD = 1            # create_dataset.py:25-27: word2id['<unk>'] = 0, word2id['<pad>'] = 1


def collate(batch):
    batch = sorted(batch, key=lambda x: np.array(x[0][0]).shape[0], reverse=True)                   # :64
    sentences = pad_sequence([torch.LongTensor(sample[0][0]) for sample in batch], padding_value=PAD)  # :70
    visual = pad_sequence([torch.FloatTensor(sample[0][1]) for sample in batch])                       # :71
    acoustic = pad_sequence([torch.FloatTensor(sample[0][2]) for sample in batch])                     # :72
    labels, emo_labels, ids = [], [], []
    for sample in batch:                                                                                # :82-93
        ids.append(sample[2])
        if sample[1].all() == 0.:
            labels.append([sample[1]][0][0])
        else:
            labels.append([np.nan_to_num(sample[1])][0][0])
    if labels[0].size == 7:                                                                             # :94-107
        labels = np.array(labels)
        filter_label = labels[:, 1:]
        for i in range(filter_label.shape[0]):
            emo_label = np.zeros(6, dtype=np.float32)
            for j, num in enumerate(filter_label[i]):
                emo_label[j] = 1 if num > 0.0 else 0
            emo_labels.append(emo_label)
        labels = labels[:, 0]
    else:
        emo_labels = None
    labels = torch.cat([torch.FloatTensor([label]) for label in labels], dim=0)                        # :115
    emo_labels = torch.from_numpy(np.array(emo_labels)) if emo_labels is not None else None            # :116
    lengths = torch.LongTensor([sample[0][0].shape[0] for sample in batch])                            # :120
    return sentences, visual, acoustic, labels, emo_labels, lengths, ids
