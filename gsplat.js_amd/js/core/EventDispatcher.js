"use strict";
// Minimal event bus, API of src/core/EventDispatcher.ts.
class EventDispatcher {
    constructor() {
        const listeners = new Map();
        this.addEventListener = (type, listener) => {
            if (!listeners.has(type)) listeners.set(type, new Set());
            listeners.get(type).add(listener);
        };
        this.removeEventListener = (type, listener) => { if (listeners.has(type)) listeners.get(type).delete(listener); };
        this.hasEventListener = (type, listener) => listeners.has(type) && listeners.get(type).has(listener);
        this.dispatchEvent = (event) => {
            if (!listeners.has(event.type)) return;
            for (const l of Array.from(listeners.get(event.type))) l(event);
        };
    }
}
module.exports = { EventDispatcher };
