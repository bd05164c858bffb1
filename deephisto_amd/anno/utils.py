"""Annotation classes -- the part of anno/utils.py the prediction path uses (anno/utils.py:20-141):
`AnnoClass`, `AnnoDescription` with `with_known_colors`, label lookup.  The palette generator
(`with_auto_colors`, distinctipy) and the matplotlib helpers are not on the path and not mirrored."""
from __future__ import annotations

from dataclasses import dataclass


@dataclass
class AnnoClass:  # anno/utils.py:20-46
    id: int
    label: str
    alternate_labels: tuple = ()
    description: str = None
    color: tuple = None

    def __str__(self) -> str:
        label = self.label_full
        description = ", " + self.description if self.description else ""
        return f"AnnoClass [{self.id}, {label}, {self.color}{description}]"

    @property
    def label_full(self) -> str:
        if not self.alternate_labels:
            return self.label
        return self.label + " (" + ", ".join(self.alternate_labels) + ")"


class AnnoDescription:  # anno/utils.py:49-141
    def __init__(self, _anno_classes) -> None:
        self.anno_classes = _anno_classes
        self.anno_classes_dict = {c.label: c for c in _anno_classes}
        for cls in _anno_classes:
            if cls.alternate_labels:
                self.anno_classes_dict.update({alt: cls for alt in cls.alternate_labels})

    @classmethod
    def with_known_colors(cls, labels_with_color: dict) -> "AnnoDescription":
        """ids follow the dict order (anno/utils.py:63-79)."""
        return AnnoDescription([AnnoClass(id=i, label=lbl, color=color)
                                for i, (lbl, color) in enumerate(labels_with_color.items())])

    def color_by_label(self, label: str):
        return self.anno_classes_dict[label].color
